#!/usr/bin/env python3
"""MI355X counterpart of the reference CLI ``inference_am_vocoder_joint.py`` (reference lines 40-156).

Same flow -- read ``<speaker>|<prompt>|<phoneme>|<content>`` lines, map tokens / speakers through the vocabulary
files, run the generator, write 16-kHz int16 wavs as ``<out>/<i+1>.wav`` -- with three differences:
  * the generator is ``JETSGeneratorHIP`` (libevhip.so) instead of the PyTorch ``JETSGenerator``;
  * lines are synthesised in batches (``--batch``) with per-utterance B=1 semantics instead of one call per line;
  * the SimBERT style encoder is out of the hot-path scope: embeddings come from ``--embeddings`` (an .npz with
    ``style`` / ``content`` arrays, one row per line) or from the deterministic placeholder of text_io.HashStyleEmbedder.

    python -m emotivoice_amd.inference_am_vocoder_joint -t data/inference/text --tokenlist .../tokenlist \
        --speakers .../speaker2 --checkpoint g_00140000 -o out_dir          (or --synthetic-weights)
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .config import EVShapes, load_yaml
from .generator import JETSGeneratorHIP
from .text_io import HashStyleEmbedder, phonemes_to_ids, read_table, read_text_file, wav_float_to_int16, write_wav_int16


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-t", "--test_file", required=True)
    ap.add_argument("--tokenlist", required=True)
    ap.add_argument("--speakers", required=True)
    ap.add_argument("-c", "--config", default=None, help="reference-format config.yaml (default: built-in joint config)")
    ap.add_argument("--checkpoint", default=None, help="generator checkpoint (torch.save dict with key 'generator')")
    ap.add_argument("--synthetic-weights", action="store_true", help="seeded synthetic checkpoint (no real weights offline)")
    ap.add_argument("--embeddings", default=None, help=".npz with 'style' and 'content' (n_lines, 768) arrays")
    ap.add_argument("-o", "--out_dir", default="test_audio")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args(argv)

    conf = load_yaml(args.config) if args.config else None
    token2id, speaker2id = read_table(args.tokenlist), read_table(args.speakers)
    gen = JETSGeneratorHIP(conf if conf is not None else EVShapes(n_vocab=len(token2id), n_speaker=len(speaker2id)))
    if args.synthetic_weights:
        from .synthetic import synth_state_dict
        sd = synth_state_dict(0, "parity", gen.shapes)
    else:
        if not args.checkpoint:
            ap.error("--checkpoint or --synthetic-weights is required")
        import torch
        sd = torch.load(args.checkpoint, map_location="cpu")["generator"]
    gen.to(args.device).load_state_dict(sd)
    gen.eval()

    lines = read_text_file(args.test_file)
    emb = np.load(args.embeddings) if args.embeddings else None
    embedder = HashStyleEmbedder(gen.shapes.bert_dim)
    os.makedirs(args.out_dir, exist_ok=True)
    todo = []
    for i, ln in enumerate(lines):
        if ln.speaker not in speaker2id:      # the reference silently skips unknown speakers (:109-110)
            continue
        ids = phonemes_to_ids(ln.phonemes, token2id)
        style = emb["style"][i] if emb is not None else embedder(ln.prompt)
        content = emb["content"][i] if emb is not None else embedder(ln.content)
        todo.append((i, dict(ling=ids, speaker=speaker2id[ln.speaker], style=style, content=content)))
    eng = gen._ensure_engine()
    written = 0
    for s in range(0, len(todo), args.batch):
        chunk = todo[s:s + args.batch]
        out = eng.synthesize([u for _, u in chunk])
        for (i, _), wav in zip(chunk, out["wav_list"]):
            write_wav_int16(os.path.join(args.out_dir, "%d.wav" % (i + 1)), wav_float_to_int16(wav), gen.shapes.sr)
            written += 1
    print("wrote %d wav files to %s" % (written, args.out_dir))
    return written


if __name__ == "__main__":
    main()
