"""`wenet.cli.reverb` of the reference -> reverb_amd.reverb."""
from reverb_amd.reverb import *  # noqa: F401,F403
from reverb_amd.reverb import ReverbASR, get_output, load_model, get_available_models, download_model  # noqa: F401
