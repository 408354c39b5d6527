"""Driver of tests/test_bench_ranks.py: runs bench.main with a host-side stand-in for WhisperHipBackend, so that what
bench.py does around the hot path -- start N ranks when asked for --gpus N, the process group, the barrier-bracketed
timed region, the one gather, the JSON line -- runs on the CPU (gloo).  It is started exactly like bench.py
(`python tests/bench_rank_driver.py --gpus 2 ...`), so the rank launcher re-starts THIS file in every rank."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from whisperx_mlx_amd import weights  # noqa: E402
from whisperx_mlx_amd.tokenizer import get_tokenizer  # noqa: E402


class StandInBackend:
    def __init__(self, model, **kw):
        self.dims = weights.MODEL_DIMS[model]
        self.tokenizer = get_tokenizer(self.dims.n_vocab)
        self.engine = types.SimpleNamespace()
        self.suppress = []
        self.stage_ms = None
        self.passes_in_flight = 1

    def _default_lanes(self, rows=16):
        return 1

    def transcribe_batch(self, segments, batch_size=16, forced_len=0, return_chunks=False, **kw):
        chunks = []
        for i, s in enumerate(segments):
            j = int(round(s["start"] / 30.0))
            toks = [self.tokenizer.timestamp_begin] + [1000 + (j * 7 + k) % 4000 for k in range(forced_len - 1)]
            chunks.append({"segment": i, "tokens": toks, "sum_logprob": -float(j), "no_speech_prob": 0.0,
                           "words": [{"word": " a", "start": 0.0, "end": 0.5}], "word_token_counts": [1]})
        if self.stage_ms is not None:
            self.stage_ms["decode"] = self.stage_ms.get("decode", 0.0) + 1.0
        return {"segments": [], "language": "en", "chunks": chunks}


if __name__ == "__main__":
    bench.main(make_backend=StandInBackend)
