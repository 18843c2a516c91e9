#!/usr/bin/env python3
"""MI355X counterpart of the reference CLI ``inference_am_vocoder_joint.py`` (reference lines 40-156).

Same command line, same directory layout, same flow:

    python -m emotivoice_amd.inference_am_vocoder_joint -d prompt_tts_open_source_joint -c config/joint \
        --checkpoint g_00140000 -t /abs/path/data/inference/text

  * ``-c/--config_folder`` is appended to ``sys.path`` and ``from config import Config`` gives the paths, exactly as the
    reference does (:151-155): ``output_directory``, ``model_config_path`` (the yacs YAML, read here with pyyaml),
    ``token_list_path``, ``speaker2id_path``, ``n_symbols`` / ``speaker_n_labels``, ``sampling_rate``;
  * checkpoints are the files of ``<output_directory>/<logdir>/ckpt`` (all of them, or the one named by ``--checkpoint``),
    each a ``torch.save`` dict with key ``generator`` (:44-73);
  * ``-t/--test_file`` holds ``<speaker>|<prompt>|<phoneme>|<content>`` lines (:96-102); unknown speakers are skipped,
    unknown phonemes raise KeyError (:109-113);
  * wavs go to ``<output_directory>/<logdir>/test_audio/audio/<checkpoint>/<i+1>.wav`` (old files there are removed first,
    :86-89), 16-bit PCM at ``config.sampling_rate`` (:130-134).
What differs: the generator is ``JETSGeneratorHIP`` (libevhip.so) and lines are synthesised in batches (``--batch``) with
per-utterance B = 1 semantics instead of one call per line.  The SimBERT style encoder (:25-38,61-67) runs on the device
(``StyleEncoderHIP``) when ``config.style_encoder_ckpt`` and the tokenizer files of ``config.bert_path`` exist on local disk; its
weights are downloads that are not part of this repo, so otherwise embeddings come from ``--embeddings`` (an .npz with ``style`` /
``content`` rows, e.g. produced by the reference's StyleEncoder) or from the deterministic placeholder of text_io.HashStyleEmbedder.
Extra flags (not in the reference): ``--embeddings --batch --device --precision --synthetic-weights``.
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

import numpy as np

from .config import load_yaml
from .generator import JETSGeneratorHIP
from .text_io import HashStyleEmbedder, phonemes_to_ids, read_table, read_text_file, wav_float_to_int16, write_wav_int16


def _load_config(config_folder: str):
    """reference :151-155: sys.path.append(<cwd>/<config_folder>); from config import Config; Config()."""
    path = config_folder if os.path.isabs(config_folder) else os.path.join(os.path.dirname(os.path.abspath("__file__")), config_folder)
    sys.modules.pop("config", None)
    sys.path.insert(0, path)            # (the reference appends; first place here so that a second call with another folder wins)
    try:
        from config import Config       # the caller's config folder, like the reference
    finally:
        sys.path.remove(path)
    return Config()


def _device_style_embedder(config, gen):
    """The reference's get_style_embedding (:25-38) with the StyleEncoder on the device: needs the style-encoder checkpoint
    (config.style_encoder_ckpt, :61-67) and the tokenizer files of config.bert_path on local disk (no hub access).  Returns a
    ``text -> (768,)`` callable, or None when either is missing (the caller falls back to --embeddings / the placeholder)."""
    ckpt, bert_path = getattr(config, "style_encoder_ckpt", None), getattr(config, "bert_path", None)
    if not ckpt or not os.path.exists(ckpt) or not bert_path or not os.path.isdir(bert_path):
        return None
    try:
        import torch

        from .simbert import StyleEncoderHIP
        from .wordpiece import load_tokenizer
        tokenizer = load_tokenizer(bert_path)             # :83 (vocab.txt -> the native WordPiece; else transformers' AutoTokenizer)
        model_ckpt = {k[7:]: v for k, v in torch.load(ckpt, map_location="cpu")["model"].items()}   # :63-66
        enc = StyleEncoderHIP(config, engine=gen._ensure_engine()).load_state_dict(model_ckpt, strict=False).eval()
    except Exception as e:                                  # noqa: BLE001 -- any missing piece means "no SimBERT available here"
        print("style encoder not available (%s): using --embeddings / the placeholder" % e)
        return None

    def embed(text):
        t = tokenizer([text], return_tensors="np")
        out = enc(input_ids=t["input_ids"], token_type_ids=t["token_type_ids"], attention_mask=t["attention_mask"])
        return np.asarray(out["pooled_output"], np.float32).squeeze()
    return embed


def synthesize_checkpoint(config, conf, checkpoint_path, out_dir, args, state_dict=None):
    token2id = read_table(config.token_list_path)            # :76-77
    speaker2id = read_table(config.speaker2id_path)          # :79-80
    gen = JETSGeneratorHIP(conf, precision=args.precision)
    if state_dict is None:
        import torch
        state_dict = torch.load(checkpoint_path, map_location="cpu")["generator"]      # :72-73
    gen.to(args.device).load_state_dict(state_dict)
    gen.eval()
    if os.path.exists(out_dir):                               # :86-89
        for j in glob.glob(os.path.join(out_dir, "*")):
            os.remove(j)
    lines = read_text_file(args.test_file)                    # :96-102
    emb = np.load(args.embeddings) if args.embeddings else None
    embedder = _device_style_embedder(config, gen) if emb is None else None
    if embedder is None:
        embedder = HashStyleEmbedder(gen.shapes.bert_dim)
    todo = []
    for i, ln in enumerate(lines):
        if ln.speaker not in speaker2id:                      # silently skipped, :109-110
            continue
        ids = phonemes_to_ids(ln.phonemes, token2id)          # KeyError on an unknown phoneme, :113
        style = emb["style"][i] if emb is not None else embedder(ln.prompt)
        content = emb["content"][i] if emb is not None else embedder(ln.content)
        todo.append((i, dict(ling=ids, speaker=speaker2id[ln.speaker], style=style, content=content)))
    eng = gen._ensure_engine()
    sr = int(getattr(config, "sampling_rate", gen.shapes.sr))
    written = 0
    for s in range(0, len(todo), args.batch):
        chunk = todo[s:s + args.batch]
        out = eng.synthesize([u for _, u in chunk])
        os.makedirs(out_dir, exist_ok=True)
        for (i, _), wav in zip(chunk, out["wav_list"]):
            write_wav_int16(os.path.join(out_dir, "%d.wav" % (i + 1)), wav_float_to_int16(wav), sr)      # :130-134
            written += 1
    gen.close()
    return written


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-d", "--logdir", type=str, required=True)
    p.add_argument("-c", "--config_folder", type=str, required=True)
    p.add_argument("--checkpoint", type=str, required=False, default="", help="inference specific checkpoint, e.g --checkpoint checkpoint_230000")
    p.add_argument("-t", "--test_file", type=str, required=True, help="the absolute path of test file that is going to inference")
    # not in the reference:
    p.add_argument("--embeddings", default=None, help=".npz with 'style' and 'content' (n_lines, 768) arrays (the SimBERT pooled outputs)")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--precision", default="mx", choices=["mx", "fast", "strict"],
                   help="mx: the contract mode (waveform within 1e-3 of the reference, fp4 cross terms); fast: fp16; strict: split precision")
    p.add_argument("--synthetic-weights", action="store_true",
                   help="no checkpoint directory: synthesise with the seeded synthetic checkpoint (named 'synthetic')")
    args = p.parse_args(argv)

    config = _load_config(args.config_folder)
    root_path = os.path.join(config.output_directory, args.logdir)          # :42
    ckpt_path = os.path.join(root_path, "ckpt")
    conf = load_yaml(config.model_config_path, n_vocab=config.n_symbols, n_speaker=config.speaker_n_labels)     # :53-58
    total = 0
    if args.synthetic_weights:
        from .config import from_reference_config
        from .synthetic import synth_state_dict
        sd = synth_state_dict(0, "parity", from_reference_config(conf))
        total += synthesize_checkpoint(config, conf, None, os.path.join(root_path, "test_audio", "audio", "synthetic"), args, sd)
    else:
        for file in sorted(os.listdir(ckpt_path)):                         # :44-49
            if args.checkpoint and file != args.checkpoint:
                continue
            total += synthesize_checkpoint(config, conf, os.path.join(ckpt_path, file),
                                           os.path.join(root_path, "test_audio", "audio", file), args)
    print("wrote %d wav files under %s" % (total, os.path.join(root_path, "test_audio", "audio")))
    return total


if __name__ == "__main__":
    print("run!")
    main()
