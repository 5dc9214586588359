// Gradient exchange entry points of the C ABI: SUM all-reduce of device ranges over an RCCL communicator created once per
// process (one process per GPU, xGMI between them).  Replaces the all-reduce torch DDP issues for the reference
// (fairseq/models/distributed_fairseq_model.py:58-67 -> torch.nn.parallel.DistributedDataParallel -> NCCL allreduce per
// 25 MB bucket): the caller hands over contiguous ranges of the flat gradient arena as the backward finishes them.
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: a process that already carries an RCCL -- PyTorch loads
// its own librccl.so -- must not get a second copy with its own allocator and IPC state, and a single-GPU process needs
// none at all.  Lookup order: the copy already loaded (RTLD_NOLOAD), $S2ST_RCCL_LIB, then librccl.so.1 / librccl.so by
// the loader's search path.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "s2st_ops.h"

namespace {

typedef struct { char internal[128]; } UniqueId;  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);

struct Api {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GetErrorStringFn error_string = nullptr;
  bool tried = false;
};
Api g_api;

bool load_api() {
  if (g_api.tried) return g_api.all_reduce != nullptr;
  g_api.tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)
    if (!g_api.lib) g_api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  if (!g_api.lib && s2st_env_str("S2ST_RCCL_LIB")) g_api.lib = dlopen(s2st_env_str("S2ST_RCCL_LIB"), RTLD_NOW | RTLD_GLOBAL);
  for (const char* n : names)
    if (!g_api.lib) g_api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!g_api.lib) return false;
  g_api.get_unique_id = (GetUniqueIdFn)dlsym(g_api.lib, "ncclGetUniqueId");
  g_api.comm_init_rank = (CommInitRankFn)dlsym(g_api.lib, "ncclCommInitRank");
  g_api.comm_destroy = (CommDestroyFn)dlsym(g_api.lib, "ncclCommDestroy");
  g_api.all_reduce = (AllReduceFn)dlsym(g_api.lib, "ncclAllReduce");
  g_api.error_string = (GetErrorStringFn)dlsym(g_api.lib, "ncclGetErrorString");
  if (!g_api.get_unique_id || !g_api.comm_init_rank || !g_api.comm_destroy || !g_api.all_reduce) {
    g_api.all_reduce = nullptr;
    return false;
  }
  return true;
}

int fail(const char* what, int rc) {
  fprintf(stderr, "[s2st] %s failed: %s (%d)\n", what, g_api.error_string ? g_api.error_string(rc) : "rccl error", rc);
  return S2ST_ERR_COMM;
}

}  // namespace

struct s2st_comm {
  Comm comm;
  int world, rank;
};

extern "C" {

// 1 if an RCCL library can be bound in this process
int s2st_comm_available(void) { return load_api() ? 1 : 0; }

// rank 0 creates the 128-byte id and hands it to the other ranks out of band (torch.distributed store, MPI, a file)
int s2st_comm_unique_id(void* id128) {
  if (!id128) return S2ST_ERR_ARG;
  if (!load_api()) return S2ST_ERR_COMM;
  UniqueId id;
  int rc = g_api.get_unique_id(&id);
  if (rc) return fail("ncclGetUniqueId", rc);
  memcpy(id128, &id, sizeof(id));
  return 0;
}

// collective over all ranks; binds the communicator to the calling thread's current HIP device
int s2st_comm_init(const void* id128, int32_t world, int32_t rank, s2st_comm** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return S2ST_ERR_ARG;
  if (!load_api()) return S2ST_ERR_COMM;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm c = nullptr;
  int rc = g_api.comm_init_rank(&c, world, id, rank);
  if (rc) return fail("ncclCommInitRank", rc);
  *out = new s2st_comm{c, world, rank};
  return 0;
}

// buf[0..n) <- sum over ranks, in place, ordered on `stream`
int s2st_allreduce_sum_f32(s2st_comm* c, float* buf, int64_t n, void* stream) {
  if (!c || (!buf && n > 0) || n < 0) return S2ST_ERR_ARG;
  if (n == 0) return 0;
  const int kFloat32 = 7, kSum = 0;  // ncclFloat32, ncclSum (rccl.h)
  int rc = g_api.all_reduce(buf, buf, (size_t)n, kFloat32, kSum, c->comm, (hipStream_t)stream);
  return rc ? fail("ncclAllReduce", rc) : 0;
}

int s2st_comm_destroy(s2st_comm* c) {
  if (!c) return 0;
  int rc = g_api.comm_destroy(c->comm);
  delete c;
  return rc ? fail("ncclCommDestroy", rc) : 0;
}

}  // extern "C"
