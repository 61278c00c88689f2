// Collectives behind the C-ABI: the cross-process forms of monica's two merges, over RCCL.
//
//   mnc_allreduce_counts     <- Counter.update of the per-sample count tables
//                               (monica/genomes/aligner.py:286-298, alignment_update) when the reads
//                               of one batch are sharded over the GPUs of a node
//   mnc_allgather_summaries  <- the hits carried between index parts
//                               (aligner.py:91-103, 196-203, 218-223) when the parts live on
//                               different GPUs: every rank needs every part's per-read summary
//
// The library does not link RCCL: librccl.so is opened on first use (a caller that never asks for
// a collective does not need it), and the communicator is the caller's -- an ncclComm_t made with
// ncclCommInitRank, or with the three helpers below, which exist so that a host program in any
// language can make one through this ABI alone.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>

#include "common.h"

namespace {

struct UniqueId { char internal[128]; };                 // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)

struct Rccl {
	void *lib = nullptr;
	int (*GetUniqueId)(UniqueId*) = nullptr;
	int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
	int (*CommDestroy)(void*) = nullptr;
	int (*CommCount)(const void*, int*) = nullptr;
	int (*AllReduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	bool ok = false;
	std::string why;                                       // what dlopen / dlsym said when it failed
};

Rccl &rccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		for (const char *name : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
			r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (r.lib) break;
			const char *err = dlerror();                       // read once, here: a second call returns NULL
			if (!r.why.empty()) r.why += "; ";
			r.why += err ? err : name;
		}
		if (!r.lib) return;
		r.why.clear();
		r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
		r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
		r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
		r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
		r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
		r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
		r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
		r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather;
		if (!r.ok) r.why = "a symbol of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclAllGather is missing";
	});
	return r;
}

int need_rccl()
{
	if (rccl().ok) return MNC_OK;
	mnc::set_error("librccl.so cannot be used: %s", rccl().why.c_str());
	return MNC_ERR_NODEVICE;
}

int check(int rc, const char *what)
{
	if (rc == 0) return MNC_OK;
	mnc::set_error("%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
	return MNC_ERR_HIP;
}

constexpr int NCCL_SUM = 0, NCCL_INT8 = 0, NCCL_INT64 = 4;

} // namespace

extern "C" int mnc_comm_unique_id(void *id128)
{
	if (!id128) return MNC_ERR_ARG;
	if (int rc = need_rccl()) return rc;
	UniqueId id;
	if (int rc = check(rccl().GetUniqueId(&id), "ncclGetUniqueId")) return rc;
	memcpy(id128, id.internal, sizeof(id.internal));
	return MNC_OK;
}

extern "C" int mnc_comm_init_rank(const void *id128, int n_ranks, int rank, void **comm)
{
	if (!id128 || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return MNC_ERR_ARG;
	if (int rc = need_rccl()) return rc;
	UniqueId id;
	memcpy(id.internal, id128, sizeof(id.internal));
	*comm = nullptr;
	return check(rccl().CommInitRank(comm, n_ranks, id, rank), "ncclCommInitRank");
}

extern "C" int mnc_comm_destroy(void *comm)
{
	if (!comm) return MNC_OK;
	if (int rc = need_rccl()) return rc;
	return check(rccl().CommDestroy(comm), "ncclCommDestroy");
}

extern "C" int mnc_comm_count(void *comm, int *n_ranks)
{
	if (!comm || !n_ranks) return MNC_ERR_ARG;
	if (int rc = need_rccl()) return rc;
	if (!rccl().CommCount) { mnc::set_error("librccl.so has no ncclCommCount"); return MNC_ERR_UNSUPPORTED; }
	return check(rccl().CommCount(comm, n_ranks), "ncclCommCount");
}

extern "C" int mnc_allreduce_counts(int64_t *d_counts, int n, void *comm, void *stream)
{
	if (!d_counts || n < 0 || !comm) return MNC_ERR_ARG;
	if (n == 0) return MNC_OK;
	if (int rc = need_rccl()) return rc;
	return check(rccl().AllReduce(d_counts, d_counts, (size_t)n, NCCL_INT64, NCCL_SUM, comm, stream), "ncclAllReduce");
}

extern "C" int mnc_allgather_summaries(const void *d_send, void *d_recv, size_t bytes_per_rank, void *comm, void *stream)
{
	if (!d_send || !d_recv || !comm) return MNC_ERR_ARG;
	if (bytes_per_rank == 0) return MNC_OK;
	if (int rc = need_rccl()) return rc;
	return check(rccl().AllGather(d_send, d_recv, bytes_per_rank, NCCL_INT8, comm, stream), "ncclAllGather");
}
