// comm.cpp — data-parallel exchange: ONE all-reduce of the flat adapter-gradient buffer per
// optimiser step, RCCL over xGMI, on the caller's HIP stream.  One process per GPU.
//
// The reference has no collective at all (single process, gradient accumulation:
// /root/reference/src/models/biomedclip/finetune.py:287-302); summing per-rank gradients of
// (local mean loss / world) is the same arithmetic as its accumulation loop.
// The 128-byte unique id is created on rank 0 and distributed by the host (torch.distributed
// broadcast / a file): this library does no rendezvous of its own.
#include <rccl/rccl.h>
#include <string.h>
#include "uia_kernels.h"

void uia_set_error(const char* fmt, ...);

static ncclComm_t g_comm = nullptr;
static int g_world = 1, g_rank = 0;

#define NCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t _r = (expr);                                                              \
        if (_r != ncclSuccess) {                                                               \
            uia_set_error("rank %d/%d: %s failed: %s", g_rank, g_world, #expr, ncclGetErrorString(_r)); \
            return -3;                                                                         \
        }                                                                                      \
    } while (0)

extern "C" {

int uia_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int uia_comm_get_unique_id(void* out, int bytes) {
    if (!out || bytes < (int)sizeof(ncclUniqueId)) { uia_set_error("uia_comm_get_unique_id: buffer too small"); return -1; }
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return 0;
}

int uia_comm_init(int rank, int world, const void* unique_id, int bytes) {
    if (g_comm) { uia_set_error("uia_comm_init: communicator already initialised"); return -1; }
    if (!unique_id || bytes < (int)sizeof(ncclUniqueId) || rank < 0 || rank >= world) { uia_set_error("uia_comm_init: bad arguments"); return -1; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    g_rank = rank;
    g_world = world;
    ncclResult_t r = ncclCommInitRank(&g_comm, world, id, rank);
    if (r != ncclSuccess) {
        uia_set_error("rank %d/%d: ncclCommInitRank failed: %s", rank, world, ncclGetErrorString(r));
        g_comm = nullptr; g_world = 1; g_rank = 0;
        return -3;
    }
    return 0;
}

int uia_comm_world(void) { return g_comm ? g_world : 1; }
int uia_comm_initialised(void) { return g_comm ? 1 : 0; }

int uia_allreduce_sum(void* stream, int dtype, void* buf, size_t n) {
    if (!g_comm) { uia_set_error("uia_allreduce_sum: communicator not initialised"); return -1; }
    if (!buf || n == 0) { uia_set_error("uia_allreduce_sum: empty buffer"); return -1; }
    const ncclDataType_t t = dtype == UIA_BF16 ? ncclBfloat16 : ncclFloat32;
    NCCL_TRY(ncclAllReduce(buf, buf, n, t, ncclSum, g_comm, (hipStream_t)stream));
    return 0;
}

int uia_allgather(void* stream, int dtype, const void* send, void* recv, size_t n_per_rank) {
    if (!g_comm) { uia_set_error("uia_allgather: communicator not initialised"); return -1; }
    if (!send || !recv || n_per_rank == 0) { uia_set_error("uia_allgather: empty buffer"); return -1; }
    const ncclDataType_t t = dtype == UIA_BF16 ? ncclBfloat16 : ncclFloat32;
    NCCL_TRY(ncclAllGather(send, recv, n_per_rank, t, g_comm, (hipStream_t)stream));
    return 0;
}

int uia_comm_destroy(void) {
    if (g_comm) { NCCL_TRY(ncclCommDestroy(g_comm)); g_comm = nullptr; g_world = 1; g_rank = 0; }
    return 0;
}

}  // extern "C"
