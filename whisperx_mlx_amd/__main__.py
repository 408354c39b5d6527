"""`python -m whisperx_mlx_amd audio.wav --model large-v3 --model_dir /ckpt -f all` (whisperx/__main__.py)."""
from .transcribe import cli

if __name__ == "__main__":
    cli()
