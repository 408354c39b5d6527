"""The synthetic audio generators live in the package (whisperx_mlx_amd/synth.py: bench.py must not depend on tests/)."""
from whisperx_mlx_amd.synth import speechlike_audio, synth_audio  # noqa: F401
