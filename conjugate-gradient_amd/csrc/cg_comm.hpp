// RCCL plumbing of the row-block multi-GPU solve: one process per GPU, rank r
// owns rows [r*G/P, (r+1)*G/P), neighbours are r-1 (above) and r+1 (below).
//
// Replaces, for the row direction, init_mpi_settings (utils.c:383-420),
// exchange_boundaries (utils.c:231-248: Isend/Irecv of the first and last row)
// and the MPI_Allreduce of one double in dot (utils.c:380).
//
// librccl is opened with dlopen on first use, so a single-GPU process never
// loads it and the library has no link-time dependency on it.  Inside a
// PyTorch process the already loaded librccl.so.1 is the one that is found.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <cstdio>
#include <cstring>

namespace cg {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    void *handle = nullptr;

    // Returns nullptr and fills msg when the library or a symbol is missing.
    static RcclApi *get(char *msg, size_t n)
    {
        static RcclApi api;
        static bool tried = false, ok = false;
        static char why[256] = "";
        if (!tried) {
            tried = true;
            const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char *nm : names) {
                api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
                if (api.handle) break;
            }
            if (!api.handle) {
                snprintf(why, sizeof why, "cannot load librccl: %s", dlerror());
            } else {
                ok = true;
                auto sym = [&](const char *s) -> void * {
                    void *p = dlsym(api.handle, s);
                    if (!p && ok) { ok = false; snprintf(why, sizeof why, "librccl lacks %s", s); }
                    return p;
                };
                api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
                api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
                api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
                api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
                api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
                api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
                api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
                api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
                api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
            }
        }
        if (!ok) { snprintf(msg, n, "%s", why); return nullptr; }
        return &api;
    }
};

class Comm {
public:
    static int unique_id(void *id128, char *msg, size_t n)
    {
        static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is expected to be 128 bytes");
        RcclApi *api = RcclApi::get(msg, n);
        if (!api) return -1;
        ncclUniqueId id;
        const ncclResult_t r = api->GetUniqueId(&id);
        if (r != ncclSuccess) { snprintf(msg, n, "ncclGetUniqueId: %s", api->GetErrorString(r)); return -1; }
        memcpy(id128, &id, sizeof id);
        return 0;
    }

    // Collective over all ranks; the current HIP device must be this rank's GPU.
    static Comm *create(const void *id128, int rank, int nranks, bool has_top, bool has_bot, char *msg, size_t n)
    {
        RcclApi *api = RcclApi::get(msg, n);
        if (!api) return nullptr;
        Comm *c = new Comm;
        c->api_ = api; c->rank_ = rank; c->nranks_ = nranks; c->has_top_ = has_top; c->has_bot_ = has_bot;
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        const ncclResult_t r = api->CommInitRank(&c->comm_, nranks, id, rank);
        if (r != ncclSuccess) {
            snprintf(msg, n, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, api->GetErrorString(r));
            c->comm_ = nullptr;
            delete c;
            return nullptr;
        }
        if (hipMalloc(&c->gather_, (size_t)nranks * sizeof(double)) != hipSuccess ||
            hipMemset(c->gather_, 0, (size_t)nranks * sizeof(double)) != hipSuccess) {
            snprintf(msg, n, "hipMalloc of the all-gather buffer failed");
            delete c;
            return nullptr;
        }
        return c;
    }

    ~Comm()
    {
        if (comm_) (void)api_->CommDestroy(comm_);
        if (gather_) (void)hipFree(gather_);
    }

    const char *error() const { return err_; }
    double *gather_buf() { return gather_; }

    // In-place all-gather of one double per rank: slot `rank` of gather_buf()
    // goes out, all slots come back.  Stream-ordered (graph-capturable).
    int allgather_scalar(hipStream_t s)
    {
        return check(api_->AllGather(gather_ + rank_, gather_, 1, ncclDouble, comm_, s), "ncclAllGather");
    }

    // Send the block's first row up and last row down; receive the neighbours'
    // boundary rows into top_dst / bot_dst.  One grouped RCCL launch.
    int exchange_rows(hipStream_t s, const double *base, long rows, long G, double *top_dst, double *bot_dst)
    {
        if (!has_top_ && !has_bot_) return 0;
        int rc = check(api_->GroupStart(), "ncclGroupStart");
        if (rc == 0 && has_bot_) {
            rc = check(api_->Send(base + (rows - 1) * G, (size_t)G, ncclDouble, rank_ + 1, comm_, s), "ncclSend(down)");
            if (rc == 0) rc = check(api_->Recv(bot_dst, (size_t)G, ncclDouble, rank_ + 1, comm_, s), "ncclRecv(from below)");
        }
        if (rc == 0 && has_top_) {
            rc = check(api_->Send(base, (size_t)G, ncclDouble, rank_ - 1, comm_, s), "ncclSend(up)");
            if (rc == 0) rc = check(api_->Recv(top_dst, (size_t)G, ncclDouble, rank_ - 1, comm_, s), "ncclRecv(from above)");
        }
        const int rc2 = check(api_->GroupEnd(), "ncclGroupEnd");
        return rc != 0 ? rc : rc2;
    }

private:
    int check(ncclResult_t r, const char *what)
    {
        if (r == ncclSuccess) return 0;
        snprintf(err_, sizeof err_, "%s: %s", what, api_->GetErrorString(r));
        return -1;
    }
    RcclApi *api_ = nullptr;
    ncclComm_t comm_ = nullptr;
    double *gather_ = nullptr;
    int rank_ = 0, nranks_ = 1;
    bool has_top_ = false, has_bot_ = false;
    char err_[256] = "";
};

} // namespace cg
