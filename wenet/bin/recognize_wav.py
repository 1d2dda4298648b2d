"""`wenet.bin.recognize_wav:main` is the reference's `reverb` console entry (pyproject.toml:31-32)."""
from reverb_amd.bin.recognize_wav import get_args, main  # noqa: F401

if __name__ == "__main__":
    main()
