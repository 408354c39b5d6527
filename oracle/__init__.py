"""CPU oracle for the WhisperX hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``whisperx_mlx_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker, never as the thing measured or shipped.

Pinning status (see DESIGN.md "Oracle"):

* ``logmel``      -- pinned: golden vectors generated from the reference's own
                     ``whisperx/audio.py:112-159`` (tests/golden/logmel_*.npz).
* ``ctc``         -- pinned: golden vectors generated from the reference's own
                     ``whisperx/alignment.py:387-613`` (tests/golden/ctc_*.npz,
                     align_*.json).
* ``whisper_ref``, ``decoding``, ``dtw`` -- PARITY UNPINNED at the reference
                     boundary: the arithmetic lives in third-party
                     ``mlx-whisper`` (branch ``whisperx-optimizations`` of
                     github.com/sooth/mlx-whisper, on ``mlx>=0.26.0``; neither
                     is vendored or installable here).  The restatement follows
                     the published OpenAI Whisper algorithm and is cross-checked
                     against HuggingFace ``transformers`` (a third-party
                     secondary oracle) with seeded random weights.
* ``quant``        -- PARITY UNPINNED: int8 weight quantisation restated from the
                     reference's unwired sketch ``backends/mlx_quantization.py``
                     (no vector, never called there).
* ``wav2vec2_ref`` -- architecture cross-checked against ``transformers``
                     ``Wav2Vec2ForCTC`` (what ``alignment.py:97-106`` loads).
"""
