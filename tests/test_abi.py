"""CPU checks of the drop-in boundary: libevhip.so loads, exports every symbol include/*.h declares,
struct sizes agree between C and ctypes, and the product path fails loudly without a HIP device."""
import ctypes as C
import glob
import os
import re
import subprocess

import pytest

from conftest import ROOT

from emotivoice_amd import _ffi


def _declared_symbols():
    syms = set()
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(hdr).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms |= set(re.findall(r"\b(ev_[a-z0-9_]+)\s*\(", src))
    return syms


def test_library_exports_every_declared_symbol():
    lib = _ffi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 19
    for s in declared:
        assert hasattr(lib, s), s
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)


def test_struct_layouts_match_the_c_headers(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "evhip.h"\n#include "evhip_ops.h"\n'
                   '#include <stddef.h>\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ev_config), sizeof(ev_result), sizeof(ev_kernel_stat), '
                   'sizeof(ev_conv_gemm_desc), sizeof(ev_res_pair_desc), offsetof(ev_res_pair_desc, epi), offsetof(ev_conv_gemm_desc, add16_a), '
                   'offsetof(ev_config, vocoder_precision));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(_ffi.ev_config), C.sizeof(_ffi.ev_result), C.sizeof(_ffi.ev_kernel_stat), C.sizeof(_ffi.ev_conv_gemm_desc),
                     C.sizeof(_ffi.ev_res_pair_desc), _ffi.ev_res_pair_desc.epi.offset, _ffi.ev_conv_gemm_desc.add16_a.offset,
                     _ffi.ev_config.vocoder_precision.offset]


def test_default_config_matches_reference_yaml_values():
    cfg = _ffi.ev_config()
    _ffi.lib().ev_default_config(C.byref(cfg))
    assert (cfg.n_vocab, cfg.n_speaker, cfg.n_mels, cfg.hidden, cfg.heads) == (502, 2014, 80, 384, 8)
    assert list(cfg.up_rates)[:4] == [8, 8, 2, 2] and list(cfg.up_kernels)[:4] == [16, 16, 4, 4]
    assert list(cfg.rb_kernels)[:3] == [3, 7, 11] and [list(r)[:3] for r in cfg.rb_dils][:3] == [[1, 3, 5]] * 3


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from emotivoice_amd.engine import EVEngine, EVError
    with pytest.raises(EVError, match="no HIP device|no CPU fallback"):
        EVEngine()
