#!/usr/bin/env python
"""Same command line as /root/reference/diarization/infer_pyannote3.0.py:16-42, on the MI355X-native
pipeline:  python -m reverb_amd.bin.infer_pyannote3 a.wav b.wav --out-dir out --pipeline-model DIR

`--pipeline-model` is a local directory (config.yaml + segmentation.pt + embedding.pt); the hub names the
reference accepts are resolved under $REVERB_DIARIZATION_HOME if set (no network I/O is done here).
"""
import argparse
import os
from pathlib import Path

from reverb_amd.diarization import Pipeline


def main(argv=None):
    parser = argparse.ArgumentParser(description='Run inference on audio files')
    parser.add_argument('audios', nargs='+')
    parser.add_argument('--out-dir', type=Path, required=True)
    parser.add_argument('--hf-access-token', type=str, required=False, default=None)
    parser.add_argument('--pipeline-model', type=str, required=False, default='Revai/reverb-diarization-v1')
    parser.add_argument('--dtype', choices=['bf16', 'f32'], default='bf16')
    args = parser.parse_args(argv)
    os.makedirs(args.out_dir, exist_ok=True)

    model = args.pipeline_model
    home = os.environ.get('REVERB_DIARIZATION_HOME')
    if not os.path.isdir(model) and home and os.path.isdir(os.path.join(home, model)):
        model = os.path.join(home, model)
    pipeline = Pipeline.from_pretrained(model, use_auth_token=args.hf_access_token, dtype=args.dtype)
    pipeline.to('cuda')

    for audio in args.audios:
        print('Processing', audio)
        annotation = pipeline(audio)
        with open(args.out_dir / f'{os.path.splitext(os.path.basename(audio))[0]}.rttm', 'w') as f:
            annotation.write_rttm(f)


if __name__ == '__main__':
    main()
