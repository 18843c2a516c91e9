"""Object wrapper over the libevhip.so handle: numpy / raw-pointer in, numpy out.  No torch needed."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _ffi
from .config import EVShapes


class EVError(RuntimeError):
    pass


_PREC = {"f16": _ffi.EV_PREC_F16, "f32": _ffi.EV_PREC_F32, "x3": _ffi.EV_PREC_X3, "mx": _ffi.EV_PREC_MX}


def resolve_precision(precision, decoder_precision, vocoder_precision):
    """``precision`` is the one-knob form: "fast" = fp16 MFMA operands on the frame-rate path (BASELINE.json's bf16 / fp16
    configs; 2.4e-3 on zero-mean waveforms), "strict" = split precision (three fp16 MFMAs per product on hi/lo parts, fp32
    activations: fp32-class accuracy, ~1e-6 relative L2 on the waveform), "mx" = the contract mode: the strict data flow, but the
    generator's layers with >= 128 channels evaluate a product as one fp16 MFMA + two block-scaled fp4 MFMAs for the cross terms
    (waveform ~4e-4 from the reference, inside north_star's 1e-3, at half the matrix work of strict).
    ``decoder_precision`` / ``vocoder_precision`` override it per component.  ``None`` everywhere = "mx", which is also what
    ``ev_default_config`` hands a C caller (ABI 5): "fast" and "strict" are explicit opt-ins."""
    if precision not in (None, "fast", "strict", "mx"):
        raise ValueError("precision must be 'fast', 'strict' or 'mx'")
    base = {None: "mx", "mx": "mx", "strict": "x3", "fast": "f16"}[precision]      # no argument = ev_default_config's own default = the contract mode
    return decoder_precision or base, vocoder_precision or base


def make_ev_config(shapes: EVShapes, decoder_precision: str = "mx", keep_stages: bool = False,
                   token_rate: str = "split", vocoder_chunk_mb: int = 0, vocoder_streams: int = 0,
                   vocoder_precision: str = "mx", mx_residual: str = "planes", decoder_attention: str = "split",
                   fused_pairs: bool = True, mx_mrf: str = "planes", decoder_ln: str = "planes", token_splitk: bool = True,
                   mx_act_format: str = "e5m2", mx_group: bool = True) -> _ffi.ev_config:
    cfg = _ffi.ev_config()
    _ffi.lib().ev_default_config(C.byref(cfg))
    for f in ("n_vocab", "n_speaker", "n_mels", "hidden", "heads", "enc_layers", "dec_layers", "ffn_kernel", "bert_dim",
              "dur_layers", "pitch_layers", "energy_layers", "var_kernel", "var_embed_kernel", "up_init_ch"):
        setattr(cfg, f, int(getattr(shapes, f)))
    cfg.n_up = len(shapes.up_rates)
    for i, (u, k) in enumerate(zip(shapes.up_rates, shapes.up_kernels)):
        cfg.up_rates[i], cfg.up_kernels[i] = int(u), int(k)
    cfg.n_rb = len(shapes.rb_kernels)
    cfg.n_rb_dils = len(shapes.rb_dils[0])
    for j, k in enumerate(shapes.rb_kernels):
        cfg.rb_kernels[j] = int(k)
        for d, v in enumerate(shapes.rb_dils[j]):
            cfg.rb_dils[j][d] = int(v)
    cfg.sample_rate = int(shapes.sr)
    cfg.decoder_precision = _PREC[decoder_precision]
    if vocoder_precision not in ("f16", "x3", "mx"):
        raise ValueError("vocoder_precision must be 'f16', 'x3' or 'mx'")
    cfg.vocoder_precision = _PREC[vocoder_precision]
    cfg.keep_stages = 1 if keep_stages else 0
    cfg.vocoder_chunk_mb = int(vocoder_chunk_mb)
    cfg.vocoder_streams = int(vocoder_streams)
    cfg.token_rate_split = {"split": 1, "f32": 0}[token_rate]
    # engine switches (ev_config; they were environment variables until round 3)
    cfg.mx_residual = {"planes": 0, "fp32": 1}[mx_residual]
    cfg.decoder_attention = {"split": 0, "f32": 1}[decoder_attention]
    cfg.fused_pairs = 0 if fused_pairs else 1
    cfg.mx_mrf = {"planes": 0, "fp32": 1}[mx_mrf]                    # running MRF sum of an MX stage: partial plane sets / an fp32 tensor
    cfg.decoder_ln_planes = {"planes": 0, "fp32": 1}[decoder_ln]     # MX decoder: LayerNorm writes its consumer's plane set / fp32 + a planes pass
    cfg.token_splitk = 0 if token_splitk else 1                       # the token-rate conv-FFN's second conv split-K (shape rule) / one pass
    cfg.mx_act_format = {"e5m2": 0, "fp4": 1}[mx_act_format]          # activation operand of the cross terms where a kernel offers both (fused C = 32 pairs)
    cfg.mx_group = 0 if mx_group else 1                               # same-level convs of a stage's three ResBlocks as one grouped launch / one launch per conv
    return cfg


class EVEngine:
    """One handle = one GPU + one stream + one workspace (include/evhip.h).  Not thread-safe."""

    def __init__(self, shapes: Optional[EVShapes] = None, device_id: int = 0, decoder_precision: Optional[str] = None,
                 keep_stages: bool = False, token_rate: str = "split", vocoder_chunk_mb: int = 0,
                 vocoder_streams: int = 0, vocoder_precision: Optional[str] = None, precision: Optional[str] = None,
                 mx_residual: str = "planes", decoder_attention: str = "split", fused_pairs: bool = True, mx_mrf: str = "planes",
                 decoder_ln: str = "planes", token_splitk: bool = True, mx_act_format: str = "e5m2", mx_group: bool = True):
        self.shapes = shapes or EVShapes()
        self._lib = _ffi.lib()
        self._h = C.c_void_p()
        decoder_precision, vocoder_precision = resolve_precision(precision, decoder_precision, vocoder_precision)
        self.decoder_precision, self.vocoder_precision = decoder_precision, vocoder_precision
        cfg = make_ev_config(self.shapes, decoder_precision, keep_stages, token_rate, vocoder_chunk_mb, vocoder_streams,
                             vocoder_precision, mx_residual, decoder_attention, fused_pairs, mx_mrf, decoder_ln, token_splitk, mx_act_format, mx_group)
        if self._lib.ev_create(device_id, C.byref(cfg), C.byref(self._h)) != 0:
            raise EVError(self._lib.ev_last_error(None).decode())
        self.device_id = device_id
        self._blob_keepalive = None
        self.last: Optional[_ffi.ev_result] = None

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ev_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EVError(self._lib.ev_last_error(self._h).decode())

    def set_stream(self, hip_stream_ptr: int):
        self._check(self._lib.ev_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def set_profiling(self, on: bool):
        self._check(self._lib.ev_set_profiling(self._h, 1 if on else 0))

    # -- weights
    def load_blob(self, blob: bytes, manifest_json: Optional[str] = None):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self._check(self._lib.ev_load_weights(self._h, C.cast(buf, C.c_void_p), len(blob),
                                              manifest_json.encode() if manifest_json else None))

    def load_blob_device(self, dptr: int, nbytes: int, keepalive=None):
        """Borrow a blob that already lives in device memory (e.g. after an RCCL broadcast)."""
        self._blob_keepalive = keepalive
        self._check(self._lib.ev_load_weights_device(self._h, C.c_void_p(dptr), nbytes, None))

    # -- SimBERT prompt / content encoder (ev_style_*)
    def style_load(self, blob: bytes, cfg: dict):
        """blob / cfg from packer.pack_bert_state_dict."""
        bc = _ffi.ev_bert_config()
        self._lib.ev_default_bert_config(C.byref(bc))
        for k in ("vocab_size", "hidden", "layers", "intermediate", "max_position", "type_vocab"):
            setattr(bc, k, int(cfg[k]))
        bc.heads = int(cfg.get("heads", bc.hidden // 64))
        bc.ln_eps = float(cfg.get("ln_eps", 1e-12))
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self._check(self._lib.ev_style_load_weights(self._h, C.byref(bc), C.cast(buf, C.c_void_p), len(blob)))
        self.style_hidden = int(bc.hidden)

    def style_embed(self, id_lists: Sequence[np.ndarray], type_lists: Optional[Sequence[np.ndarray]] = None) -> np.ndarray:
        """pooled_output (B, hidden) of B token-id sequences (one text each: [CLS] ... [SEP])."""
        B = len(id_lists)
        ids = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int64).reshape(-1) for x in id_lists]))
        cu = np.zeros(B + 1, np.int32)
        cu[1:] = np.cumsum([len(x) for x in id_lists])
        tt = None
        if type_lists is not None:
            tt = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int64).reshape(-1) for x in type_lists]))
        out = np.empty((B, self.style_hidden), np.float32)
        self._check(self._lib.ev_style_embed(self._h, B, ids.ctypes.data_as(C.c_void_p), tt.ctypes.data_as(C.c_void_p) if tt is not None else None,
                                             cu.ctypes.data_as(C.c_void_p), 0, out.ctypes.data_as(C.c_void_p)))
        return out

    # -- raw calls (host or device pointers)
    def synthesize_raw(self, B: int, ling_ptr: int, cu_seqlens: np.ndarray, speaker_ptr: int, style_ptr: int,
                       content_ptr: int, alpha: float = 1.0, flags: int = 0) -> _ffi.ev_result:
        cu = np.ascontiguousarray(cu_seqlens, np.int32)
        res = _ffi.ev_result()
        self._check(self._lib.ev_synthesize(self._h, B, C.c_void_p(ling_ptr), cu.ctypes.data_as(C.c_void_p),
                                            C.c_void_p(speaker_ptr), C.c_void_p(style_ptr), C.c_void_p(content_ptr),
                                            C.c_float(alpha), flags, C.byref(res)))
        self.last = res
        return res

    def vocoder_raw(self, B: int, mel_ptr: int, mel_is_f16: bool, mel_lens: np.ndarray, flags: int = 0) -> _ffi.ev_result:
        ml = np.ascontiguousarray(mel_lens, np.int32)
        res = _ffi.ev_result()
        self._check(self._lib.ev_vocoder(self._h, B, C.c_void_p(mel_ptr), 1 if mel_is_f16 else 0,
                                         ml.ctypes.data_as(C.c_void_p), flags, C.byref(res)))
        self.last = res
        return res

    def set_forced_durations(self, durations: np.ndarray):
        d = np.ascontiguousarray(durations, np.int64)
        self._check(self._lib.ev_set_forced_durations(self._h, d.ctypes.data_as(C.c_void_p), d.size))

    # -- helpers
    def d2h(self, dev_ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        if out.nbytes:
            self._check(self._lib.ev_memcpy_d2h(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(dev_ptr), out.nbytes))
        return out

    def result_to_numpy(self, res: _ffi.ev_result, want_int16: bool = False) -> Dict[str, object]:
        B = res.batch
        mel_lens = np.array([res.mel_lens[b] for b in range(B)], np.int32)
        mel_offs = np.array([res.mel_offsets[b] for b in range(B + 1)], np.int64)
        up = self.shapes.upsample_factor
        out: Dict[str, object] = dict(mel_lens=mel_lens, mel_offsets=mel_offs)
        if res.wav:
            out["wav"] = self.d2h(res.wav, (res.total_samples,), np.float32)
            out["wav_list"] = [out["wav"][mel_offs[b] * up:mel_offs[b + 1] * up] for b in range(B)]
        if want_int16 and res.wav_i16:
            out["wav_i16"] = self.d2h(res.wav_i16, (res.total_samples,), np.int16)
        if res.mel:
            out["mel"] = self.d2h(res.mel, (res.total_frames, self.shapes.n_mels), np.float32)
            out["mel_list"] = [out["mel"][mel_offs[b]:mel_offs[b + 1]] for b in range(B)]
        if res.durations:
            out["durations"] = self.d2h(res.durations, (res.total_tokens,), np.int64)
            out["log_durations"] = self.d2h(res.log_durations, (res.total_tokens,), np.float32)
            out["pitch"] = self.d2h(res.pitch, (res.total_tokens,), np.float32)
            out["energy"] = self.d2h(res.energy, (res.total_tokens,), np.float32)
        return out

    # -- numpy convenience API
    def synthesize(self, utts: Sequence[dict], alpha: float = 1.0, want_int16: bool = False, vocoder: bool = True,
                   forced_durations: Optional[np.ndarray] = None) -> Dict[str, object]:
        """utts: dicts with ling (N,) int64, speaker int, style (768,), content (768,) -- the four fields the
        reference builds per input line (inference_am_vocoder_joint.py:113-119)."""
        B = len(utts)
        ling = np.ascontiguousarray(np.concatenate([np.asarray(u["ling"], np.int64) for u in utts]))
        cu = np.zeros(B + 1, np.int32)
        cu[1:] = np.cumsum([len(u["ling"]) for u in utts])
        spk = np.ascontiguousarray([int(u["speaker"]) for u in utts], np.int64)
        style = np.ascontiguousarray(np.stack([np.asarray(u["style"], np.float32) for u in utts]))
        content = np.ascontiguousarray(np.stack([np.asarray(u["content"], np.float32) for u in utts]))
        flags = 0
        if want_int16:
            flags |= _ffi.EV_FLAG_WANT_INT16
        if not vocoder:
            flags |= _ffi.EV_FLAG_NO_VOCODER
        if forced_durations is not None:
            self.set_forced_durations(forced_durations)
            flags |= _ffi.EV_FLAG_FORCED_DURATIONS
        res = self.synthesize_raw(B, ling.ctypes.data, cu, spk.ctypes.data, style.ctypes.data, content.ctypes.data, alpha, flags)
        out = self.result_to_numpy(res, want_int16)
        out["cu_seqlens"] = cu
        return out

    def vocoder(self, mels: Sequence[np.ndarray], want_int16: bool = False) -> Dict[str, object]:
        """mels: list of (n_mels, T_b) arrays (the reference's (B,80,T) layout per utterance), fp32 or fp16."""
        is16 = mels[0].dtype == np.float16
        flat = np.ascontiguousarray(np.concatenate([np.ascontiguousarray(m, mels[0].dtype).ravel() for m in mels]))
        lens = np.array([m.shape[1] for m in mels], np.int32)
        flags = _ffi.EV_FLAG_WANT_INT16 if want_int16 else 0
        res = self.vocoder_raw(len(mels), flat.ctypes.data, is16, lens, flags)
        return self.result_to_numpy(res, want_int16)

    # receptive field of the generator in mel frames per side: conv_post 3 samples -> 60-sample ResBlock halos per stage through the
    # four transposed convs -> 11 frames, + conv_pre 3 = 14 (derivation in DESIGN.md section 4); 16 is used
    VOCODER_CONTEXT_FRAMES = 16

    def vocoder_chunked(self, mel: np.ndarray, chunk_frames: int = 256, context: Optional[int] = None):
        """Streaming vocoding of one long mel (n_mels, T): yields the waveform of consecutive chunks of ``chunk_frames`` frames.
        Every chunk is vocoded with ``context`` extra frames on each side and the centre is kept; because every output sample
        only depends on +-14 mel frames and the kernels are position-independent, the concatenation is BIT-IDENTICAL to vocoding
        the whole mel at once (tests/test_gpu_parity.py).  Bounds the vocoder workspace for arbitrarily long utterances and gives
        first audio after one chunk (ROADMAP "Support longer text")."""
        ctx = self.VOCODER_CONTEXT_FRAMES if context is None else context
        up = self.shapes.upsample_factor
        T = mel.shape[1]
        for a in range(0, T, chunk_frames):
            b = min(T, a + chunk_frames)
            lo, hi = max(0, a - ctx), min(T, b + ctx)
            wav = self.vocoder([np.ascontiguousarray(mel[:, lo:hi])])["wav"]
            yield wav[(a - lo) * up:(b - lo) * up]

    def get_stage(self, name: str) -> np.ndarray:
        """Stage tap of the last call (SURVEY.md Appendix C names; needs keep_stages=True): (rows, C) fp32."""
        need = self._lib.ev_get_stage(self._h, name.encode(), None, 0)
        if need < 0:
            raise EVError(self._lib.ev_last_error(self._h).decode())
        if name in ("dur", "mel_len"):
            out = np.empty(need // 8, np.int64)
        else:
            out = np.empty(need // 4, np.float32)
        got = self._lib.ev_get_stage(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes)
        if got < 0:
            raise EVError(self._lib.ev_last_error(self._h).decode())
        return out

    def timings(self) -> Dict[str, float]:
        out = {}
        ms = C.c_float()
        for name in ("total", "am", "encoder", "variance", "decoder", "vocoder"):
            if self._lib.ev_get_timing(self._h, name.encode(), C.byref(ms)) == 0:
                out[name] = float(ms.value)
        return out

    def launch_records(self) -> List[dict]:
        """Per-launch records of the last profiled call (set_profiling(True)), in launch order."""
        out = []
        r = _ffi.ev_launch_record()
        for i in range(self._lib.ev_launch_record_count(self._h)):
            self._lib.ev_get_launch_record(self._h, i, C.byref(r))
            out.append(dict(name=r.name.decode(), M=r.M, N=r.N, K=r.K, taps=r.taps, dil=r.dil, ms=float(r.ms), flops=float(r.flops), bytes=float(r.bytes)))
        return out

    def kernel_stats(self) -> List[dict]:
        out = []
        st = _ffi.ev_kernel_stat()
        for i in range(self._lib.ev_kernel_stat_count(self._h)):
            self._lib.ev_get_kernel_stat(self._h, i, C.byref(st))
            out.append(dict(name=st.name.decode(), launches=st.launches, ms=float(st.ms), flops=float(st.flops), bytes=float(st.bytes)))
        return out
