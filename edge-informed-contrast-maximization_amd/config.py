"""Plain-PyYAML reader for the reference's hydra config tree (hydra / omegaconf are not installable offline).

Supports what src/experiments/e00/configs uses: a ``defaults`` list of config groups (``- dataset: dsec``), absolute
``${a.b.c}`` interpolation, the ``divide`` resolver registered at src/experiments/e00/__main__.py:20-22, and
``key=value`` / ``group=name`` command-line overrides (run.sh).  Returns nested ``AttrDict``s.
"""
import os
import re

import yaml

_INTERP = re.compile(r'\$\{([^${}]+)\}')


class _Loader(yaml.SafeLoader):
    """SafeLoader that, like OmegaConf's, reads ``1e-7`` as a float (YAML 1.1 wants ``1.0e-7``); the reference writes
    ``gtol: 1e-7`` (configs/main.yaml:41)."""


_Loader.add_implicit_resolver(
    'tag:yaml.org,2002:float',
    re.compile(r'''^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+]?[0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$''', re.X),
    list('-+0123456789.'))


def _yaml_load(text):
    return yaml.load(text, Loader=_Loader)


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return AttrDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def _get(root, path):
    cur = root
    for p in path.split('.'):
        cur = cur[int(p)] if isinstance(cur, list) else cur[p]
    return cur


def _set(root, path, value):
    parts = path.split('.')
    cur = root
    for p in parts[:-1]:
        if p not in cur or not isinstance(cur[p], dict):
            cur[p] = {}
        cur = cur[p]
    cur[parts[-1]] = value


def _resolve_value(root, v, depth=0):
    if depth > 32:
        raise ValueError('interpolation cycle')
    if isinstance(v, dict):
        return {k: _resolve_value(root, x, depth) for k, x in v.items()}
    if isinstance(v, list):
        return [_resolve_value(root, x, depth) for x in v]
    if not isinstance(v, str) or '${' not in v:
        return v

    def one(expr):
        expr = expr.strip()
        if expr.startswith('divide:'):
            a, b = (s.strip() for s in expr[len('divide:'):].split(',', 1))
            fa = _resolve_value(root, a if a.startswith('${') else _coerce(a), depth + 1)
            fb = _resolve_value(root, b if b.startswith('${') else _coerce(b), depth + 1)
            return fa / fb                      # __main__.py:21 (true division)
        return _resolve_value(root, _get(root, expr), depth + 1)

    # innermost-first so that ${divide:${a.b},5} works
    while True:
        m = _INTERP.search(v)
        if m is None:
            return _coerce(v) if isinstance(v, str) else v
        val = one(m.group(1))
        if m.start() == 0 and m.end() == len(v):
            return val
        v = v[:m.start()] + str(val) + v[m.end():]


def _coerce(s):
    try:
        return _yaml_load(s)
    except Exception:
        return s


def load_config(config_dir, config_name='main', overrides=()):
    """Compose ``<config_dir>/<config_name>.yaml`` with its defaults and overrides, resolve interpolations."""
    with open(os.path.join(config_dir, config_name + '.yaml')) as f:
        main = _yaml_load(f.read()) or {}
    defaults = main.pop('defaults', [])
    groups = {}
    for d in defaults:
        if isinstance(d, dict):
            groups.update(d)
    plain = []
    for ov in overrides:
        k, _, val = ov.partition('=')
        k = k.lstrip('+')
        if k in groups and os.path.isdir(os.path.join(config_dir, k)):
            groups[k] = val
        else:
            plain.append((k, _coerce(val)))
    cfg = {}
    for g, name in groups.items():
        if name is None:
            continue
        cfg[g] = _load_group(config_dir, g, name)
    for k, v in main.items():          # `_self_` first in the reference: main's own keys, then groups; no key overlaps
        cfg[k] = v
    for k, v in plain:
        _set(cfg, k, v)
    return _wrap(_resolve_value(cfg, cfg))


def _load_group(config_dir, group, name):
    path = os.path.join(config_dir, group, str(name) + '.yaml')
    with open(path) as f:
        node = _yaml_load(f.read()) or {}
    sub_defaults = node.pop('defaults', []) if isinstance(node, dict) else []
    for d in sub_defaults:              # nested groups, e.g. edge_extraction/default.yaml -> clahe: default
        if isinstance(d, dict):
            for g, n in d.items():
                if n is not None and os.path.isdir(os.path.join(config_dir, group, g)):
                    node[g] = _load_group(config_dir, os.path.join(group, g), n)
    return node
