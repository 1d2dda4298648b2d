"""`reverb` console script: same flags and output files as the reference's
`asr/wenet/bin/recognize_wav.py` (:29-147 flags, :150-208 main): one `<result_dir>/<mode>/<audio>.ctm`
per decoding mode.  Extra flags: --dtype, --max_chunks."""
from __future__ import annotations

import argparse
import logging
import os
from pathlib import Path

MODES = ["attention", "ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring", "joint_decoding"]


def get_args(argv=None):
    p = argparse.ArgumentParser(description="recognize with your model")
    p.add_argument("--model", default=None, help="reverb model name or a directory with config.yaml and a .pt file")
    p.add_argument("--config", default=None, help="config file")
    p.add_argument("--checkpoint", default=None, help="checkpoint model")
    p.add_argument("--audio_file", required=True, help="audio to transcribe")
    p.add_argument("--gpu", type=int, default=-1, help="gpu id for this rank, -1 means device 0")
    p.add_argument("--tokenizer-symbols", help="Path to tk.units.txt. Overrides the config path.")
    p.add_argument("--bpe-path", help="Path to tk.model. Overrides the config path.")
    p.add_argument("--cmvn-path", help="Path to cmvn. Overrides the config path.")
    p.add_argument("--beam_size", type=int, default=10, help="beam size for search (CTC modes: at most 16, the device top-k width)")
    p.add_argument("--length_penalty", type=float, default=0.0, help="length penalty")
    p.add_argument("--blank_penalty", type=float, default=0.0, help="blank penalty")
    p.add_argument("--result_dir", required=True, help="asr result file")
    p.add_argument("--batch_size", type=int, default=1, help="batch size")
    p.add_argument("--chunk_size", type=int, default=2051, help="Chunk size")
    p.add_argument("--modes", nargs="+", default=["attention_rescoring"], choices=MODES,
                   help="One or more supported decoding mode.")
    p.add_argument("--ctc_weight", type=float, default=0.1, help="ctc weight for rescoring weight in attention rescoring")
    p.add_argument("--decoding_chunk_size", type=int, default=-1, help="decoding chunk size (<0: full chunk)")
    p.add_argument("--num_decoding_left_chunks", type=int, default=-1, help="number of left chunks for decoding")
    p.add_argument("--simulate_streaming", action="store_true", help="simulate streaming inference")
    p.add_argument("--reverse_weight", type=float, default=0.0, help="right to left weight for attention rescoring")
    p.add_argument("--overwrite_cmvn", action="store_true", help="overwrite CMVN params in model with those in configs")
    p.add_argument("--verbatimicity", type=float, default=1.0, help="the level of verbatimicity to run the model")
    p.add_argument("--timings_adjustment", type=float, default=230, help="time shift applied to all timings (ms)")
    p.add_argument("--log_level", default="INFO", help="log level")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="device compute mode")
    p.add_argument("--max_chunks", type=int, default=64, help="chunks per device batch")
    return p.parse_args(argv)


def main(argv=None):
    args = get_args(argv)
    logging.basicConfig(level=getattr(logging, str(args.log_level).upper(), logging.INFO),
                        format="%(asctime)s %(levelname)s %(message)s")
    from reverb_amd.reverb import ReverbASR, load_model
    model_set = args.model is not None
    pair_set = args.checkpoint is not None and args.config is not None
    if model_set == pair_set:
        raise RuntimeError("One of either --model or (--checkpoint and --config) must be set.")
    if args.model:
        reverb = load_model(args.model, gpu=args.gpu, dtype=args.dtype, max_chunks=args.max_chunks)
    else:
        reverb = ReverbASR(args.config, args.checkpoint, cmvn_path=args.cmvn_path,
                           tokenizer_symbols=args.tokenizer_symbols, bpe_path=args.bpe_path, gpu=args.gpu,
                           overwrite_cmvn=args.overwrite_cmvn, dtype=args.dtype, max_chunks=args.max_chunks)
    outputs = reverb.transcribe_modes(
        args.audio_file, args.modes, format="ctm", verbatimicity=args.verbatimicity, chunk_size=args.chunk_size,
        batch_size=args.batch_size, beam_size=args.beam_size, decoding_chunk_size=args.decoding_chunk_size,
        num_decoding_left_chunks=args.num_decoding_left_chunks, ctc_weight=args.ctc_weight,
        simulate_streaming=args.simulate_streaming, reverse_weight=args.reverse_weight,
        blank_penalty=args.blank_penalty, length_penalty=args.length_penalty,
        timings_adjustment=args.timings_adjustment)
    stem = Path(args.audio_file).with_suffix(".ctm").name
    for mode, text in zip(args.modes, outputs):
        out_dir = os.path.join(args.result_dir, mode)
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, stem), "w", encoding="utf-8") as f:
            f.write(text)
        logging.info("wrote %s", os.path.join(out_dir, stem))


if __name__ == "__main__":
    main()
