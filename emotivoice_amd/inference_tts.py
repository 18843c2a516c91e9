#!/usr/bin/env python3
"""MI355X counterpart of the reference's bulk driver ``inference_tts.py`` (reference lines 46-230): raw text in, one wav per line out.

Same command line and the same observable behaviour:

    python -m emotivoice_amd.inference_tts -d prompt_tts_open_source_joint -c config/joint --checkpoint g_00140000 \
        -t /abs/path/texts.txt [-o out_dir] [-g 0,1,2,3] [-n 1]

  * every non-skipped line ``i`` of ``--text_file`` is ONE utterance: content = the stripped line (:121), phonemes = ``g2p(content)``
    (:122-123), prompt = ``PROMPTS[i % 4]`` and speaker id = ``i % n_speaker`` (:97-100: 'Happy', 'Excited', 'Sad', 'Angry'; all speakers
    of ``config.speaker2id_path`` in file order);
  * outputs: ``<output_dir>/<speaker name>/<i+1:06d>.wav`` (16-bit PCM at ``config.sampling_rate``, ``* MAX_WAV_VALUE`` then an int16 C
    cast, :145-150) and ``<...>.txt`` holding the content line (:151-153); ``output_dir`` defaults to
    ``<config.output_directory>/<logdir>/audio`` (:53-57);
  * resume: a line whose wav already exists is skipped before any work is done for it (:106-109);
  * a line that fails (unknown phoneme, G2P error, ...) prints ``Error: ...`` and is skipped, the run goes on (:154-156);
  * the lines are cut into ``gpus x num_thread`` contiguous chunks, the first ``total % n`` chunks one line longer (:197-222), chunk ``j`` runs
    on GPU ``j % gpus`` in its own process.

What differs: each process synthesises its chunk in BATCHES of ``--batch`` lines on ``JETSGeneratorHIP``'s engine (per-utterance results do
not depend on the batch: the engine's batch-invariance tests) instead of one ``generator(...)`` call per line; the four prompt embeddings are
computed once per process; G2P is a callable -- ``--g2p pkg.module:function`` or, by default, the reference's own ``frontend.g2p_cn_en`` if
that module is importable (jieba / pypinyin / g2p_en are host packages outside this repo), run over a fork pool (``FrontendPool``) when
``--frontend_workers`` > 1 -- and ``--num_thread`` > 1 only adds more chunks per GPU (the reference used it to hide its CPU front end behind
several 4-GB model replicas; one engine saturates the device here).  The SimBERT encoder runs on the device when
``config.style_encoder_ckpt`` / ``config.bert_path`` are on local disk, otherwise the documented placeholder embedder is used.
Extra flags (not in the reference): ``--batch --precision --g2p --frontend_workers --synthetic-weights --phoneme-input``.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .text_io import HashStyleEmbedder, phonemes_to_ids, read_table, wav_float_to_int16, write_wav_int16

PROMPTS = ["Happy", "Excited", "Sad", "Angry"]          # inference_tts.py:87


def split_chunks(total_len: int, thread_num: int) -> List[Tuple[int, int]]:
    """(start, count) of every worker, the reference's arithmetic (:197-222): ``chunk = total // n`` with the first ``total % n`` workers
    one longer; fewer lines than workers: one line each for the first ``total`` workers and EMPTY ranges past the end for the rest
    (the reference starts those processes too; they find no line in their range)."""
    if total_len >= thread_num:
        chunk_size, remains = total_len // thread_num, total_len - (total_len // thread_num) * thread_num
    else:
        chunk_size, remains = 1, 0
    out, begin = [], 0
    for _ in range(thread_num):
        n = chunk_size + (1 if remains > 0 else 0)
        remains -= 1 if remains > 0 else 0
        out.append((begin, n))
        begin += n
    return out


def utt_paths(output_dir: str, speaker_name: str, i: int) -> Tuple[str, str, str]:
    """(speaker directory, wav path, txt path) of line i (:101-106, :151)."""
    d = os.path.join(output_dir, speaker_name)
    name = "%06d" % (i + 1)
    return d, os.path.join(d, name + ".wav"), os.path.join(d, name + ".txt")


def load_g2p(spec: Optional[str]) -> Callable[[str], str]:
    """``pkg.module:function`` -> callable(content) -> space-separated phoneme string; None: the reference's frontend if importable."""
    if spec:
        mod, _, fn = spec.partition(":")
        return getattr(importlib.import_module(mod), fn or "g2p")
    try:                                                     # the reference's own front end, when the caller runs inside its tree
        from frontend import g2p_cn_en                       # type: ignore
        from frontend_en import ROOT_DIR, G2p, read_lexicon  # type: ignore
        lexicon, g2p = read_lexicon("%s/lexicon/librispeech-lexicon.txt" % ROOT_DIR), G2p()
        return lambda content: g2p_cn_en(content, g2p, lexicon)
    except Exception as e:                                   # noqa: BLE001
        raise RuntimeError("no G2P available (%s): pass --g2p pkg.module:function or --phoneme-input" % e)


def run_chunk(lines: Sequence[str], start_idx: int, chunk_num: int, *, synthesize: Callable[[List[dict]], Dict[str, object]],
              embed: Callable[[str], np.ndarray], g2p: Callable[[str], str], token2id: Dict[str, int], id2speaker: Dict[int, str],
              output_dir: str, sampling_rate: int, batch: int = 32, g2p_map: Optional[Callable[[Sequence[str]], List[str]]] = None,
              n_speaker: Optional[int] = None, log: Callable[[str], None] = print) -> Dict[str, int]:
    """Lines [start_idx, start_idx + chunk_num) of the text file -> wav + txt files (the body of the reference's ``main``, :89-156).
    ``synthesize(list of utterance dicts) -> {"wav_list": [...]}`` is EVEngine.synthesize; ``embed(text) -> (768,)``.
    Returns counts: written / skipped_existing / errors."""
    n_speaker = n_speaker or len(id2speaker)          # (:88: range(conf.n_speaker))
    prompt_emb = {p: np.asarray(embed(p), np.float32) for p in PROMPTS}          # four prompts: once per process, not once per line
    stats = dict(written=0, skipped_existing=0, errors=0)
    todo = []                                                                     # (i, content, speaker id, wav path, txt path)
    for i in range(start_idx, min(start_idx + chunk_num, len(lines))):
        speaker = i % n_speaker
        d, wav_path, txt_path = utt_paths(output_dir, id2speaker[speaker], i)
        os.makedirs(d, exist_ok=True)                                             # (:101-103: made before the resume check)
        if os.path.exists(wav_path):
            log("audio %s exists, continue." % wav_path)
            stats["skipped_existing"] += 1
            continue
        todo.append((i, lines[i].strip(), speaker, wav_path, txt_path))
    for s in range(0, len(todo), batch):
        group = todo[s:s + batch]
        contents = [g[1] for g in group]
        try:
            phones = g2p_map(contents) if g2p_map is not None else None
        except Exception:                                                         # noqa: BLE001 -- fall back to per-line G2P to isolate the bad line
            phones = None
        utts, keep = [], []
        for j, (i, content, speaker, wav_path, txt_path) in enumerate(group):
            try:
                text = (phones[j] if phones is not None else g2p(content)).split()
                ids = phonemes_to_ids(text, token2id)                             # KeyError on an unknown phoneme (:130)
                if len(ids) == 0:
                    raise ValueError("empty phoneme sequence")
                utts.append(dict(ling=ids, speaker=speaker, style=prompt_emb[PROMPTS[i % len(PROMPTS)]],
                                 content=np.asarray(embed(content), np.float32)))
                keep.append((content, wav_path, txt_path))
            except Exception as e:                                                # noqa: BLE001 -- the reference's per-line try/except (:154-156)
                log("Error: %s" % (e,))
                stats["errors"] += 1
        if not utts:
            continue
        try:
            wavs = list(synthesize(utts)["wav_list"])
        except Exception:                                                         # noqa: BLE001 -- re-run the batch a line at a time: only the
            wavs = []                                                             # offending line is lost, as in the reference (:154-156)
            for u in utts:
                try:
                    wavs.append(synthesize([u])["wav_list"][0])
                except Exception as e:                                            # noqa: BLE001
                    log("Error: %s" % (e,))
                    stats["errors"] += 1
                    wavs.append(None)
        for (content, wav_path, txt_path), wav in zip(keep, wavs):
            if wav is None:
                continue
            write_wav_int16(wav_path, wav_float_to_int16(wav), sampling_rate)     # :145-150
            with open(txt_path, "w", encoding="utf-8") as f:                      # :151-153
                f.write("%s\n" % content)
            stats["written"] += 1
    return stats


def _load_config(config_folder: str):
    from .inference_am_vocoder_joint import _load_config as load
    return load(config_folder)


def main_worker(args, config, gpu_id: int, start_idx: int, chunk_num: int, state_dict=None) -> Dict[str, int]:
    """One process of the reference (:46-88 setup + the chunk loop): generator, tables, style encoder, G2P, then run_chunk."""
    from .config import load_yaml
    from .generator import JETSGeneratorHIP
    from .inference_am_vocoder_joint import _device_style_embedder
    root_path = os.path.join(config.output_directory, args.logdir)
    output_dir = args.output_dir or os.path.join(root_path, "audio")              # :53-57
    os.makedirs(output_dir, exist_ok=True)
    conf = load_yaml(config.model_config_path, n_vocab=config.n_symbols, n_speaker=config.speaker_n_labels)     # :59-63
    # the G2P worker pool forks its workers in its constructor: it must exist BEFORE this process initialises the HIP runtime
    # (.to("cuda:N") below) -- a forked child of an initialised parent inherits runtime threads, locks and KFD state
    g2p = (lambda s: s) if args.phoneme_input else load_g2p(args.g2p)
    pool = None
    if args.frontend_workers > 1 and not args.phoneme_input:
        from .frontend_pool import FrontendPool
        pool = FrontendPool(g2p, workers=args.frontend_workers)
    gen = JETSGeneratorHIP(conf, precision=args.precision)
    if state_dict is None:
        if args.synthetic_weights:
            from .config import from_reference_config
            from .synthetic import synth_state_dict
            state_dict = synth_state_dict(0, "parity", from_reference_config(conf))
        else:
            import torch
            state_dict = torch.load(os.path.join(root_path, "ckpt", args.checkpoint), map_location="cpu")["generator"]     # :73-75
    gen.to("cuda:%d" % gpu_id).load_state_dict(state_dict)
    gen.eval()
    token2id = read_table(config.token_list_path)                                 # :78-79
    with open(config.speaker2id_path, encoding="utf-8") as f:                       # :81-82
        id2speaker = {idx: t.strip() for idx, t in enumerate(f.readlines())}
    embed = _device_style_embedder(config, gen) or HashStyleEmbedder(gen.shapes.bert_dim)
    with open(args.text_file, "r", encoding="utf-8") as f:
        lines = f.readlines()
    eng = gen._ensure_engine()
    try:
        stats = run_chunk(lines, start_idx, chunk_num, synthesize=eng.synthesize, embed=embed, g2p=g2p, token2id=token2id,
                          id2speaker=id2speaker, output_dir=output_dir, sampling_rate=int(getattr(config, "sampling_rate", gen.shapes.sr)),
                          batch=args.batch, g2p_map=pool.map if pool is not None else None, n_speaker=int(config.speaker_n_labels))
    finally:
        if pool is not None:
            pool.close()
        gen.close()
    print("part [%d, %d): %s" % (start_idx, start_idx + chunk_num, stats))
    return stats


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("-d", "--logdir", default="prompt_tts_open_source_joint", type=str, required=False)
    p.add_argument("-c", "--config_folder", default="config/joint", type=str, required=False)
    p.add_argument("--checkpoint", type=str, default="g_00140000", required=False, help="inference specific checkpoint")
    p.add_argument("-t", "--text_file", type=str, required=True, help="the absolute path of test file")
    p.add_argument("-o", "--output_dir", type=str, required=False, default=None, help="path to save the generated audios.")
    p.add_argument("-g", "--gpu_ids", type=str, required=False, default="0")
    p.add_argument("-n", "--num_thread", type=str, required=False, default="1")
    # not in the reference:
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--precision", default="mx", choices=["mx", "fast", "strict"],
                   help="mx: the contract mode (waveform within 1e-3 of the reference, fp4 cross terms); fast: fp16; strict: split precision")
    p.add_argument("--g2p", default=None, help="pkg.module:function mapping a content line to a space-separated phoneme string")
    p.add_argument("--frontend_workers", type=int, default=1, help="fork-pool workers for the G2P of a process")
    p.add_argument("--phoneme-input", action="store_true", help="the lines already are space-separated phoneme tokens")
    p.add_argument("--synthetic-weights", action="store_true", help="seeded synthetic checkpoint instead of <logdir>/ckpt/<checkpoint>")
    return p


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    config = _load_config(args.config_folder)
    gpu_list = args.gpu_ids.split(",")
    thread_num = len(gpu_list) * int(args.num_thread)                             # :189-195
    with open(args.text_file, "r", encoding="utf-8") as f:
        total_len = sum(1 for _ in f)
    print("Total texts: %d, Thread nums: %d" % (total_len, thread_num))
    parts = split_chunks(total_len, thread_num)
    if thread_num == 1:
        main_worker(args, config, int(gpu_list[0]), *parts[0])
        return 0
    import multiprocessing as mp
    ctx = mp.get_context("spawn")                 # (the HIP runtime does not survive a fork of an initialised parent)
    procs = []
    for j, (begin, n) in enumerate(parts):
        print("process part %d..." % j)
        pr = ctx.Process(target=_spawn_entry, args=(vars(args), args.config_folder, int(gpu_list[j % len(gpu_list)]), begin, n))
        pr.start()
        procs.append(pr)
    for pr in procs:                              # (the reference joins only its last process, :229-230; every one is joined here)
        pr.join()
    return 0


def _spawn_entry(arg_dict, config_folder, gpu_id, begin, n):
    main_worker(argparse.Namespace(**arg_dict), _load_config(config_folder), gpu_id, begin, n)


if __name__ == "__main__":
    sys.exit(main())
