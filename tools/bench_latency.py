import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from emotivoice_amd.engine import EVEngine
from emotivoice_amd.packer import pack_state_dict
from emotivoice_amd.synthetic import synth_inputs, synth_state_dict
blob, man = pack_state_dict(synth_state_dict(0, "bench"))
for prec in ("fast", "strict"):
    eng = EVEngine(precision=prec)
    eng.load_blob(blob, man)
    for nph in (16, 64, 256):
        u = synth_inputs(99, [nph], None)[0]
        ling1 = np.ascontiguousarray(u["ling"]); cu1 = np.array([0, nph], np.int32)
        spk1 = np.zeros(1, np.int64); st1 = np.ascontiguousarray(u["style"]); ct1 = np.ascontiguousarray(u["content"])
        best = 1e9
        for it in range(23):
            t1 = time.perf_counter()
            r1 = eng.synthesize_raw(1, ling1.ctypes.data, cu1, spk1.ctypes.data, st1.ctypes.data, ct1.ctypes.data, 1.0, 0)
            dtl = time.perf_counter() - t1
            if it >= 3: best = min(best, dtl)
        print(prec, "SMALLM=%s" % os.environ.get("EV_GEMM_SMALLM", "default"), nph, "phonemes  %.3f ms" % (best * 1e3), flush=True)
    eng.close()
