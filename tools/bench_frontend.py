#!/usr/bin/env python3
"""Front-end (G2P) throughput on the host: lines/s of FrontendPool around a G2P of realistic cost (tools/g2p_standin.py), serial and with
4 / 16 / 64 workers -- to be read beside the engine's utterances/s (reference: one line at a time inside the
synthesis process, inference_tts.py:63-71).     python tools/bench_frontend.py [--lines 20000] [--json out.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g2p_standin import make_g2p, make_lexicon, make_text  # noqa: E402

from emotivoice_amd.frontend_pool import FrontendPool  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=20000)
    ap.add_argument("--workers", default="1,4,16,64")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    t0 = time.perf_counter()
    lex = make_lexicon()
    g2p = make_g2p(lex)
    texts = make_text(lex, args.lines)
    phon = [g2p(t) for t in texts[:2000]]
    res = dict(host_cores=os.cpu_count(), lexicon_words=len(lex), lines=args.lines, setup_s=round(time.perf_counter() - t0, 2),
               phonemes_per_line=round(sum(len(p.split()) for p in phon) / len(phon), 1), runs=[])
    for w in [int(x) for x in args.workers.split(",")]:
        if w > (os.cpu_count() or 1) * 2:
            continue
        with FrontendPool(g2p, workers=w, chunk=64) as pool:
            pool.map(texts[:1000])                      # warm the workers (fork + first touch of the lexicon pages)
            best = max(pool.throughput(texts) for _ in range(3))
        res["runs"].append(dict(workers=w, lines_per_s=round(best, 1), lines_per_s_per_worker=round(best / w, 1)))
        print("workers %3d: %9.0f lines/s  (%7.0f per worker)" % (w, best, best / w), flush=True)
    print(json.dumps(res))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
