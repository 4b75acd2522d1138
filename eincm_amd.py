"""Importable alias for the package directory ``edge-informed-contrast-maximization_amd`` (a hyphenated name cannot
appear in an ``import`` statement):  ``import eincm_amd; eincm_amd.losses.loss_func(...)``."""
import importlib
import sys

_pkg = importlib.import_module('edge-informed-contrast-maximization_amd')
sys.modules[__name__] = _pkg
for _sub in ('synth', '_lib', 'engine', 'losses', 'sharding', 'solver', 'staging', 'config', 'evaluation', 'edges'):
    sys.modules[f'{__name__}.{_sub}'] = importlib.import_module(f'edge-informed-contrast-maximization_amd.{_sub}')
