"""-m gpu: wx_gather_results -- the one collective of the multi-GPU path through the C ABI, over a real RCCL
communicator (one rank on this box's GPU; the 8-GPU scaling run belongs to the driver).  The records are the
fixed-width ones parallel.pack_records builds."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from whisperx_mlx_amd import _lib, parallel as P      # noqa: E402


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_gather_results_over_rccl_one_rank():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        res = [{"tokens": [50365, 11, 12, 13, 50400], "sum_logprob": -2.5, "no_speech_prob": 0.5,
                "word_spans": [(2, 0, 300), (3, 300, 700)]},
               {"tokens": list(range(100, 130)), "sum_logprob": -9.0, "no_speech_prob": 0.0, "word_spans": []}]
        local = P.pack_records(res, [4, 9]).cuda()
        out = torch.full_like(local, -7)
        s = torch.cuda.current_stream()
        rc = _lib.lib().wx_gather_results(comm, _lib.ptr(local), local.numel() * 4, _lib.ptr(out), C.c_void_p(s.cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(out, local)
        got = P.unpack_records(out)
        assert [g["chunk_id"] for g in got] == [4, 9] and got[0]["word_spans"] == [(2, 0, 300), (3, 300, 700)]
        assert _lib.lib().wx_gather_results(None, _lib.ptr(local), 8, _lib.ptr(out), None) != 0     # no communicator: refused
    finally:
        rccl.ncclCommDestroy(comm)
