"""CPU checks of the drop-in boundary: libevhip.so loads, exports every symbol include/*.h declares,
struct sizes agree between C and ctypes, and the product path fails loudly without a HIP device."""
import ctypes as C
import glob
import os
import re
import shutil
import subprocess
import sys

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
    if shutil.which("gcc") is None:
        pytest.skip("gcc not installed")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(_ffi.ev_config), C.sizeof(_ffi.ev_result), C.sizeof(_ffi.ev_kernel_stat), C.sizeof(_ffi.ev_conv_gemm_desc),
                     C.sizeof(_ffi.ev_res_pair_desc), _ffi.ev_res_pair_desc.epi.offset, _ffi.ev_conv_gemm_desc.add16_a.offset,
                     _ffi.ev_config.vocoder_precision.offset]


def test_abi_info_reports_what_the_binding_expects():
    """ev_abi_info is what _ffi.lib() checks at load time: a stale libevhip.so or a stale binding must fail there."""
    sizes = (C.c_size_t * 4)()
    assert _ffi.lib().ev_abi_info(sizes) == _ffi.EV_ABI_VERSION == 7
    assert tuple(sizes) == (C.sizeof(_ffi.ev_config), C.sizeof(_ffi.ev_result), C.sizeof(_ffi.ev_conv_gemm_desc), C.sizeof(_ffi.ev_res_pair_desc))
    hdr = open(os.path.join(ROOT, "include", "evhip.h")).read()
    assert re.search(r"#define EV_ABI_VERSION\s+%d\b" % _ffi.EV_ABI_VERSION, hdr)


def test_engine_switches_are_config_fields_not_environment_variables():
    """Round 3's EV_MX_RESPL / EV_ATTN_F32 / EV_NO_FUSED_PAIR became ev_config fields; the product library calls getenv nowhere."""
    cfg = _ffi.ev_config()
    _ffi.lib().ev_default_config(C.byref(cfg))
    assert (cfg.mx_residual, cfg.decoder_attention, cfg.fused_pairs, cfg.mx_mrf, cfg.decoder_ln_planes, cfg.token_splitk, cfg.mx_act_format, cfg.mx_group) == (0, 0, 0, 0, 0, 0, 0, 0)
    from emotivoice_amd.engine import make_ev_config
    from emotivoice_amd.config import EVShapes
    c2 = make_ev_config(EVShapes(), "mx", vocoder_precision="mx", mx_residual="fp32", decoder_attention="f32", fused_pairs=False, mx_mrf="fp32",
                        decoder_ln="fp32", token_splitk=False, mx_act_format="fp4", mx_group=False)
    assert (c2.mx_residual, c2.decoder_attention, c2.fused_pairs, c2.mx_mrf, c2.decoder_ln_planes, c2.token_splitk, c2.mx_act_format, c2.mx_group) == (1, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(KeyError):
        make_ev_config(EVShapes(), mx_residual="bf16")
    out = subprocess.run(["nm", "-D", "--undefined-only", _ffi.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in out, "the product library must not read environment variables"


def test_default_config_matches_reference_yaml_values():
    cfg = _ffi.ev_config()
    _ffi.lib().ev_default_config(C.byref(cfg))
    assert (cfg.n_vocab, cfg.n_speaker, cfg.n_mels, cfg.hidden, cfg.heads) == (502, 2014, 80, 384, 8)
    assert list(cfg.up_rates)[:4] == [8, 8, 2, 2] and list(cfg.up_kernels)[:4] == [16, 16, 4, 4]
    assert list(cfg.rb_kernels)[:3] == [3, 7, 11] and [list(r)[:3] for r in cfg.rb_dils][:3] == [[1, 3, 5]] * 3


def test_default_config_is_the_contract_precision():
    """a caller following INTEGRATION.md literally (ev_default_config, no field overridden) must get the mode that
    meets north_star's 1e-3 on zero-mean audio -- EV_PREC_MX for both components -- in C, in make_ev_config and in EVEngine()'s resolution;
    fp16 operands (2.4e-3 on zero-mean audio) are an explicit opt-in everywhere."""
    cfg = _ffi.ev_config()
    _ffi.lib().ev_default_config(C.byref(cfg))
    assert (cfg.abi_version, cfg.decoder_precision, cfg.vocoder_precision) == (7, _ffi.EV_PREC_MX, _ffi.EV_PREC_MX)
    from emotivoice_amd.config import EVShapes
    from emotivoice_amd.engine import make_ev_config
    from emotivoice_amd.generator import DEFAULT_PRECISION
    c2 = make_ev_config(EVShapes())
    assert (c2.decoder_precision, c2.vocoder_precision) == (_ffi.EV_PREC_MX, _ffi.EV_PREC_MX) and DEFAULT_PRECISION == "mx"
    src = open(os.path.join(ROOT, "emotivoice_amd", "csrc", "ev_engine.cpp")).read()
    body = src[src.index("void ev_default_config("):src.index("int ev_abi_info(")]
    assert "EV_PREC_F16" not in body and body.count("EV_PREC_MX") == 2
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "EV_PREC_MX" in integ and "ev_default_config" in integ


def test_stale_library_fails_with_the_rebuild_hint(tmp_path):
    """a library from before ev_abi_info existed must raise the ImportError that names build.py, not a bare AttributeError
    from the symbol-binding loop.  Stand-in for the stale .so: a shared object that exports nothing of the ABI."""
    src = tmp_path / "stale.c"
    src.write_text("int ev_create(void) { return 0; }\n")
    so = tmp_path / "libstale.so"
    if shutil.which("gcc") is None:
        pytest.skip("gcc not installed")
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['EVHIP_LIB'] = %r\n"
            "from emotivoice_amd import _ffi\n"
            "try:\n    _ffi.lib()\nexcept ImportError as e:\n    assert 'build.py' in str(e) and 'ev_abi_info' in str(e), e; print('OK')\n" % (ROOT, str(so)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.stdout.strip() == "OK", r.stderr


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from emotivoice_amd.engine import EVEngine, EVError
    with pytest.raises(EVError, match="no HIP device|no CPU fallback"):
        EVEngine()


def test_ev_create_rejects_shapes_the_kernels_do_not_build():
    """Config validation happens before any device is touched, so it is testable without a GPU: shapes the kernels would silently
    mishandle must be refused with a message, not discovered as garbage audio."""
    lib = _ffi.lib()

    def create(**kw):
        cfg = _ffi.ev_config()
        lib.ev_default_config(C.byref(cfg))
        for k, v in kw.items():
            if isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    getattr(cfg, k)[i] = x
            else:
                setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.ev_create(0, C.byref(cfg), C.byref(h))
        msg = lib.ev_last_error(None).decode()
        if rc == 0:
            lib.ev_destroy(h)
        return rc, msg

    for kw, needle in ((dict(up_init_ch=1024), "upsample_initial_channel"), (dict(ffn_kernel=11), "conv kernels"),
                       (dict(var_embed_kernel=13), "conv kernels"), (dict(rb_kernels=[3, 7, 15]), "ResBlock kernel"),
                       (dict(vocoder_precision=1), "vocoder_precision"), (dict(decoder_precision=7), "decoder_precision"),
                       (dict(n_rb=5), "generator layout"), (dict(abi_version=99), "abi_version"), (dict(abi_version=1), "abi_version"),
                       (dict(mx_residual=2), "mx_residual"), (dict(mx_mrf=3), "mx_mrf"), (dict(abi_version=2), "abi_version"), (dict(abi_version=3), "abi_version"),
                       (dict(token_splitk=2), "token_splitk"), (dict(mx_act_format=2), "mx_act_format"), (dict(mx_group=2), "mx_group")):
        rc, msg = create(**kw)
        assert rc != 0 and needle in msg, (kw, msg)


def test_precision_names_resolve():
    from emotivoice_amd.engine import resolve_precision
    assert resolve_precision(None, None, None) == ("mx", "mx")       # no argument = the contract mode = ev_default_config's own default
    assert resolve_precision("mx", None, None) == ("mx", "mx")
    assert resolve_precision("fast", None, None) == ("f16", "f16")
    assert resolve_precision("strict", None, None) == ("x3", "x3")
    assert resolve_precision("strict", "f32", None) == ("f32", "x3")
    assert resolve_precision(None, None, "x3") == ("mx", "x3")
    assert resolve_precision("fast", None, "x3") == ("f16", "x3")
    with pytest.raises(ValueError):
        resolve_precision("fastest", None, None)
