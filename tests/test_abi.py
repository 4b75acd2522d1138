"""The C-ABI library: builds for gfx950 without a GPU, loads, exports every symbol include/eincm.h declares, and its
host-only entry points agree with the oracle.  No GPU compute calls here."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

from oracle import eincm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'eincm.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(eincm_[a-z_0-9]+)\s*\(', txt)))


def test_header_symbols_exported(built_lib):
    syms = _declared_symbols()
    assert 'eincm_loss_grad' in syms and 'eincm_set_windows' in syms and len(syms) >= 12
    raw = C.CDLL(os.path.join(ROOT, 'edge-informed-contrast-maximization_amd', 'libeincm_hip.so'))
    for s in syms:
        assert hasattr(raw, s), f'{s} declared in include/eincm.h but not exported'


def test_binding_table_matches_header(built_lib):
    L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
    assert sorted(n for n, _, _ in L.SIGNATURES) == _declared_symbols()
    assert built_lib.eincm_abi_version() == 6


def test_struct_layouts(built_lib, tmp_path):
    """ctypes mirrors have the size the C compiler gives the structs of include/eincm.h."""
    import subprocess
    L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include "eincm.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(eincm_params),sizeof(eincm_aux),sizeof(eincm_timings),sizeof(eincm_objectives_out),'
                   'sizeof(eincm_tiled_out));return 0;}\n')
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', str(exe), str(src)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(L.Params), C.sizeof(L.Aux), C.sizeof(L.Timings), C.sizeof(L.ObjectivesOut),
                     C.sizeof(L.TiledOut)]


@pytest.mark.parametrize('R', [1, 2, 3, 5, 8])
def test_multi_ref_weights(built_lib, R):
    eng = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
    np.testing.assert_allclose(eng.multi_ref_weights(R), O.compute_weights_for_multi_reference(R), rtol=1e-15)


@pytest.mark.parametrize('method', ['bilinear', 'lanczos3', 'lanczos5', 'cubic'])
@pytest.mark.parametrize('n_in,n_out', [(1, 260), (2, 346), (16, 260), (8, 480), (260, 260), (32, 16), (5, 7)])
def test_resample_matrix(built_lib, method, n_in, n_out):
    eng = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
    np.testing.assert_allclose(eng.resample_matrix(n_in, n_out, method), O.resample_matrix(n_in, n_out, n_out / n_in, method),
                               rtol=0, atol=1e-15)


def test_bad_arguments_rejected_without_gpu(built_lib):
    assert built_lib.eincm_multi_ref_weights(0, None) == -1
    assert built_lib.eincm_multi_ref_weights(17, None) == -1
    assert built_lib.eincm_resample_matrix(0, 4, 0, None) == -1
    assert not built_lib.eincm_create(0, 1, 1, 1, 1, 1, 0)       # H, W too small -> NULL + message
    assert b'bad argument' in built_lib.eincm_last_error(None)


def test_no_cpu_fallback(built_lib):
    """Without a GPU the engine must refuse to construct (never compute on the CPU)."""
    import torch
    if torch.cuda.is_available() or torch.cuda.device_count() > 0 or os.access('/dev/kfd', os.W_OK):
        pytest.skip('GPU present')
    eng = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
    with pytest.raises(eng.EincmError, match='no HIP device|no CPU fallback'):
        eng.Engine((64, 64), 100)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'edge-informed-contrast-maximization_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
