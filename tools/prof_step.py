"""one 16-row pass of the hot path per step variant given on the command line (for rocprofv3 --kernel-trace --stats)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd.synth import speechlike_audio
from whisperx_mlx_amd.backend import WhisperHipBackend

v = int(sys.argv[1])
be = WhisperHipBackend("large-v3", max_batch=16, random_init=True, seed=0, step_variant=v)
audio = speechlike_audio(480.0, seed=1234).reshape(16, 480000)
dev = torch.from_numpy(audio).cuda()
segs = [{"start": 0.0, "end": 30.0, "audio": dev[i]} for i in range(16)]
for _ in range(2):
    be.transcribe_batch(segs, batch_size=16, language="en", word_timestamps="dtw", forced_len=145, passes_in_flight=1)
torch.cuda.synchronize()
