// wx_gather_results (include/wxhip.h): the ONE collective of the multi-GPU path -- an all-gather of the
// fixed-width per-chunk result records over RCCL/xGMI -- for hosts that own an ncclComm_t (a C/C++ embedding of the
// library).  The Python host issues the same collective through torch.distributed (backend "nccl" = RCCL), whose
// communicator is not reachable as an ncclComm_t (whisperx_mlx_amd/parallel.py).
//
// libwxhip.so does not link RCCL: the process that calls this already has it loaded (PyTorch-ROCm ships librccl.so,
// a C++ host links it), so the symbol is taken from the running process.  A second, private copy of RCCL next to
// the host's would not share its communicators.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstddef>

#include "../../include/wxhip.h"

namespace {
// ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
//                            ncclComm_t comm, hipStream_t stream);      ncclInt8 / ncclChar == 0, ncclSuccess == 0
using all_gather_fn = int (*)(const void*, void*, size_t, int, void*, hipStream_t);

all_gather_fn find_all_gather() {
    static all_gather_fn fn = [] {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!sym) {
            // a host that loaded RCCL with RTLD_LOCAL (ctypes.CDLL's default): ask for the already-loaded library by name
            for (const char* name : {"librccl.so", "librccl.so.1"}) {
                if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) {
                    sym = dlsym(h, "ncclAllGather");
                    if (sym) break;
                }
            }
        }
        return reinterpret_cast<all_gather_fn>(sym);
    }();
    return fn;
}
}  // namespace

extern "C" int wx_gather_results(void* nccl_comm, const void* local, size_t bytes_per_rank, void* all_out, void* stream) {
    if (!nccl_comm || !local || !all_out || bytes_per_rank == 0) return -2;
    all_gather_fn ag = find_all_gather();
    if (!ag) return -4;          // no RCCL in this process: there is no fallback transport
    return ag(local, all_out, bytes_per_rank, /*ncclChar*/ 0, nccl_comm, (hipStream_t)stream) == 0 ? 0 : -1;
}
