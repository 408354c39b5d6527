"""The launcher of one engine context (a "lane" of the backend's scheduler): it enqueues the passes dealt to
its context on the context's stream -- at most two enqueued and not yet turned into text --, turns a finished pass's pinned
results into text and words while the next one runs, and reports what the context counted.  One thread per lane; the job's
description and result slots are shared by the lanes of a scheduler run (`PassJob`).

The reference has no counterpart: it decodes one VAD segment after the other on one stream
(/root/reference/whisperx/backends/mlx_lightning.py:82-119)."""
from dataclasses import dataclass
from typing import Any, Callable, List, Optional

import torch


@dataclass
class PassJob:
    """what every lane of one scheduler run needs to know (WhisperHipBackend._decode_chunks_locked builds it)"""
    passes: List[list]                  # the chunks of every pass, in launch order
    pass_start: List[int]               # index of a pass's first chunk in the (sorted) chunk list
    prompt: List[int]
    dtw: Any                            # False / "upstream" / "inrepo"
    forced_len: int
    cross_split: int
    fc2_tile_n: int
    flens: Optional[list]               # bench workload: forced length per chunk of the sorted list
    launch_rows: Callable[[int], int]   # rows a pass of n chunks is launched with (scheduler.JobPlan.launch_rows)
    language: str
    results: List[Any]                  # per pass: the list of per-chunk dicts (filled by the lanes)
    errors: List[BaseException]         # raised on the calling thread afterwards


class Lane:
    """one engine context and its launcher state: at most two passes enqueued and not yet turned into text"""

    def __init__(self, backend, eng, job: PassJob):
        self.backend, self.eng, self.job = backend, eng, job
        self.slots, self.pending, self.j, self.selfq = backend._slots(eng), [], 0, 0
        # a launcher thread stays ~32 decode steps ahead of its stream (wx_tuning.max_steps_ahead) instead of enqueueing a
        # whole pass at once; the pre-warm enqueues of the scheduler, made from the calling thread one engine after the
        # other, must not wait for the GPU and do not
        self.steps_ahead = 0

    def enqueue(self, i):
        job = self.job
        if len(self.pending) == 2:
            self.finish_one()
        slot = self.slots[self.j & 1]
        self.j += 1
        n = len(job.passes[i])
        self.backend._enqueue_pass(self.eng, slot, job.passes[i], job.prompt, job.dtw, job.forced_len, job.cross_split, job.fc2_tile_n,
                                   None if job.flens is None else job.flens[job.pass_start[i]: job.pass_start[i] + n],
                                   launch_rows=job.launch_rows(n), steps_ahead=self.steps_ahead)
        self.pending.append((i, slot))

    def finish_one(self):
        i, slot = self.pending.pop(0)
        self.job.results[i] = self.backend._finish_pass(slot, self.job.language, self.job.dtw)

    def run(self, todo, steps_ahead=0):
        torch.cuda.set_device(self.eng.device)
        self.steps_ahead = steps_ahead
        try:
            for i in todo:
                self.enqueue(i)
                if len(self.pending) == 2:       # turn the older pass into text while the newer one runs
                    self.finish_one()
            while self.pending:
                self.finish_one()
            self.eng.check_status()   # raises if a kernel's bounded wait gave up (rows would be poisoned)
            self.selfq = self.eng.decode_stats()["selfq"]
        except BaseException as e:    # noqa: BLE001 - re-raised on the calling thread
            self.job.errors.append(e)
