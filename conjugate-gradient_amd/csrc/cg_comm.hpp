// Communication of the row-block multi-GPU solve: one process per GPU, rank r
// owns rows [r*G/P, (r+1)*G/P), neighbours are r-1 (above) and r+1 (below).
//
// Replaces, for the row direction, init_mpi_settings (utils.c:383-420),
// exchange_boundaries (utils.c:231-248: Isend/Irecv of the first and last row)
// and the MPI_Allreduce of one double in dot (utils.c:380).
//
// Two transports behind one interface:
//   RcclComm  RCCL over xGMI, stream-ordered, no host involvement (the fast path).
//             librccl is opened with dlopen on first use, so a single-GPU process
//             never loads it and the library has no link-time dependency on it;
//             inside a PyTorch process the already loaded librccl.so.1 is found.
//   HostComm  device rows / scalars are staged through pinned host memory and
//             handed to two caller-supplied functions (MPI, torch.distributed
//             with gloo, ...).  Synchronous and slow; it exists so that every
//             multi-rank code path of the solver can be exercised by several
//             processes sharing ONE GPU, and as a fallback where RCCL is absent.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <cstdio>
#include <cstring>

#include "../../include/cg_hip.h"

namespace cg {

class Comm {
public:
    virtual ~Comm()
    {
        if (gather_) (void)hipFree(gather_);
        if (block_) (void)hipFree(block_);
    }
    const char *error() const { return err_; }
    // Device buffer of nranks doubles; slot `rank` is this rank's contribution.
    double *gather_buf() { return gather_; }
    // In-place all-gather of one double per rank, ordered after prior work on s.
    int allgather_scalar(hipStream_t s) { return allgather(s, gather_, 1); }
    // Device buffer of nranks * kBlockCap doubles for all-gathers of `count` doubles per rank
    // (count <= kBlockCap): rank r's contribution sits at block_buf() + r*count.
    static constexpr long kBlockCap = 8192;
    double *block_buf() { return block_; }
    int allgather_block(hipStream_t s, long count) { return allgather(s, block_, count); }
    // In-place all-gather inside `buf`: `count` doubles per rank, rank r's at buf + r*count.
    virtual int allgather(hipStream_t s, double *buf, long count) = 0;
    // Send the block's first row up and its last row down; receive the neighbours'
    // boundary rows into top_dst / bot_dst.  Ordered after prior work on s.
    virtual int exchange_rows(hipStream_t s, const double *base, long rows, long G, double *top_dst, double *bot_dst) = 0;
    // True if the calls above only enqueue stream work (hipGraph-capturable).
    virtual bool stream_ordered() const = 0;
    // True if a one-rank communicator stands in for every peer (timing rehearsal): nothing can be agreed with anyone.
    virtual bool self_only() const { return false; }
    // Self-test helper: move n doubles from src to dst through the transport's own
    // point-to-point primitives addressed to this rank itself (a no-op copy where the
    // transport has no such notion).  Lets a single process exercise send/receive.
    virtual int loopback(hipStream_t s, const double *src, double *dst, long n)
    {
        return hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -1;
    }

    // Single-rank rehearsal of the halo traffic: the block's first and last row are sent to this rank
    // itself in one grouped send/receive, into scratch rows.  Transports without such a notion refuse.
    virtual int rehearse_rows(hipStream_t, const double *, long, long)
    {
        snprintf(err_, sizeof err_, "this transport cannot rehearse the halo exchange");
        return -1;
    }
    // Allocates the scratch rows of rehearse_rows (set-up time: never under stream capture).
    virtual int prepare_rehearsal(long) { return rehearse_rows(nullptr, nullptr, 0, 0); }

protected:
    bool alloc_gather(int nranks)
    {
        const size_t blk = (size_t)nranks * kBlockCap * sizeof(double);
        return hipMalloc(&gather_, (size_t)nranks * sizeof(double)) == hipSuccess &&
               hipMemset(gather_, 0, (size_t)nranks * sizeof(double)) == hipSuccess &&
               hipMalloc(&block_, blk) == hipSuccess && hipMemset(block_, 0, blk) == hipSuccess &&
               hipStreamSynchronize(nullptr) == hipSuccess;
    }
    double *gather_ = nullptr, *block_ = nullptr;
    int rank_ = 0, nranks_ = 1;
    bool has_top_ = false, has_bot_ = false;
    char err_[256] = "";
};

// ------------------------------------------------------------------- RCCL ----

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

class RcclComm : public Comm {
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
    // self_only: a one-rank communicator stands in for every peer (timing rehearsal on one GPU): rows are
    // sent to and received from this rank itself, the all-gathers run over this rank's slice alone.
    static RcclComm *create(const void *id128, int rank, int nranks, bool has_top, bool has_bot, char *msg, size_t n,
                            bool self_only = false)
    {
        RcclApi *api = RcclApi::get(msg, n);
        if (!api) return nullptr;
        RcclComm *c = new RcclComm;
        c->api_ = api; c->rank_ = rank; c->nranks_ = nranks; c->has_top_ = has_top; c->has_bot_ = has_bot;
        c->self_only_ = self_only;
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        const ncclResult_t r = self_only ? api->CommInitRank(&c->comm_, 1, id, 0) : api->CommInitRank(&c->comm_, nranks, id, rank);
        if (r != ncclSuccess) {
            snprintf(msg, n, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, api->GetErrorString(r));
            c->comm_ = nullptr;
            delete c;
            return nullptr;
        }
        if (!c->alloc_gather(nranks)) {
            snprintf(msg, n, "hipMalloc of the all-gather buffer failed");
            delete c;
            return nullptr;
        }
        return c;
    }

    ~RcclComm() override
    {
        if (comm_) (void)api_->CommDestroy(comm_);
        if (scratch_rows_) (void)hipFree(scratch_rows_);
    }

    bool stream_ordered() const override { return true; }
    bool self_only() const override { return self_only_; }

    int allgather(hipStream_t s, double *buf, long count) override
    {
        double *mine = buf + (long)rank_ * count;
        return check(api_->AllGather(mine, self_only_ ? mine : buf, (size_t)count, ncclDouble, comm_, s), "ncclAllGather");
    }

    // One grouped RCCL launch for the (up to) two sends and two receives.
    int exchange_rows(hipStream_t s, const double *base, long rows, long G, double *top_dst, double *bot_dst) override
    {
        if (!has_top_ && !has_bot_) return 0;
        int rc = check(api_->GroupStart(), "ncclGroupStart");
        const int below = self_only_ ? 0 : rank_ + 1, above = self_only_ ? 0 : rank_ - 1;
        if (rc == 0 && has_bot_) {
            rc = check(api_->Send(base + (rows - 1) * G, (size_t)G, ncclDouble, below, comm_, s), "ncclSend(down)");
            if (rc == 0) rc = check(api_->Recv(bot_dst, (size_t)G, ncclDouble, below, comm_, s), "ncclRecv(from below)");
        }
        if (rc == 0 && has_top_) {
            rc = check(api_->Send(base, (size_t)G, ncclDouble, above, comm_, s), "ncclSend(up)");
            if (rc == 0) rc = check(api_->Recv(top_dst, (size_t)G, ncclDouble, above, comm_, s), "ncclRecv(from above)");
        }
        const int rc2 = check(api_->GroupEnd(), "ncclGroupEnd");
        return rc != 0 ? rc : rc2;
    }

    int prepare_rehearsal(long G) override
    {
        if (scratch_rows_ && scratch_len_ >= G) return 0;
        if (scratch_rows_) (void)hipFree(scratch_rows_);
        scratch_rows_ = nullptr; scratch_len_ = 0;
        if (hipMalloc(&scratch_rows_, (size_t)(2 * G) * sizeof(double)) != hipSuccess) {
            snprintf(err_, sizeof err_, "hipMalloc of the rehearsal rows failed");
            return -1;
        }
        scratch_len_ = G;
        return 0;
    }

    int rehearse_rows(hipStream_t s, const double *base, long rows, long G) override
    {
        if (!scratch_rows_ || scratch_len_ < G) {
            snprintf(err_, sizeof err_, "rehearsal rows not prepared");
            return -1;
        }
        const int self = self_only_ ? 0 : rank_;
        int rc = check(api_->GroupStart(), "ncclGroupStart");
        if (rc == 0) rc = check(api_->Send(base + (rows - 1) * G, (size_t)G, ncclDouble, self, comm_, s), "ncclSend(self, last row)");
        if (rc == 0) rc = check(api_->Recv(scratch_rows_, (size_t)G, ncclDouble, self, comm_, s), "ncclRecv(self)");
        if (rc == 0) rc = check(api_->Send(base, (size_t)G, ncclDouble, self, comm_, s), "ncclSend(self, first row)");
        if (rc == 0) rc = check(api_->Recv(scratch_rows_ + G, (size_t)G, ncclDouble, self, comm_, s), "ncclRecv(self)");
        const int rc2 = check(api_->GroupEnd(), "ncclGroupEnd");
        return rc != 0 ? rc : rc2;
    }

    int loopback(hipStream_t s, const double *src, double *dst, long n) override
    {
        int rc = check(api_->GroupStart(), "ncclGroupStart");
        const int self = self_only_ ? 0 : rank_;
        if (rc == 0) rc = check(api_->Send(src, (size_t)n, ncclDouble, self, comm_, s), "ncclSend(self)");
        if (rc == 0) rc = check(api_->Recv(dst, (size_t)n, ncclDouble, self, comm_, s), "ncclRecv(self)");
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
    bool self_only_ = false;
    double *scratch_rows_ = nullptr;      // rehearsal only
    long scratch_len_ = 0;
};

// ------------------------------------------------------------- host-staged ---

class HostComm : public Comm {
public:
    static HostComm *create(cg_exchange_fn ex, cg_allgather_fn ag, void *user, int rank, int nranks, long G,
                            bool has_top, bool has_bot, char *msg, size_t n)
    {
        HostComm *c = new HostComm;
        c->ex_ = ex; c->ag_ = ag; c->user_ = user;
        c->rank_ = rank; c->nranks_ = nranks; c->has_top_ = has_top; c->has_bot_ = has_bot; c->G_ = G;
        const size_t row = (size_t)G * sizeof(double);
        bool ok = c->alloc_gather(nranks);
        for (int i = 0; i < 4 && ok; ++i) ok = hipHostMalloc(reinterpret_cast<void **>(&c->rowbuf_[i]), row, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&c->hgather_), (size_t)nranks * kBlockCap * sizeof(double), hipHostMallocDefault) == hipSuccess;
        if (!ok) { snprintf(msg, n, "allocation of the host staging buffers failed"); delete c; return nullptr; }
        return c;
    }

    ~HostComm() override
    {
        for (double *p : rowbuf_) if (p) (void)hipHostFree(p);
        if (hgather_) (void)hipHostFree(hgather_);
    }

    bool stream_ordered() const override { return false; }

    int allgather(hipStream_t s, double *buf, long count) override
    {
        if (count < 1 || count > kBlockCap) { snprintf(err_, sizeof err_, "all-gather of %ld doubles per rank", count); return -1; }
        double *mine = hgather_ + (long)rank_ * count;
        if (hip(hipMemcpyAsync(mine, buf + (long)rank_ * count, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, s), "D2H")) return -1;
        if (hip(hipStreamSynchronize(s), "sync")) return -1;
        if (ag_(user_, mine, count, hgather_) != 0) { snprintf(err_, sizeof err_, "host all-gather callback failed"); return -1; }
        if (hip(hipMemcpyAsync(buf, hgather_, (size_t)nranks_ * count * sizeof(double), hipMemcpyHostToDevice, s), "H2D")) return -1;
        return hip(hipStreamSynchronize(s), "sync");
    }

    int exchange_rows(hipStream_t s, const double *base, long rows, long G, double *top_dst, double *bot_dst) override
    {
        if (!has_top_ && !has_bot_) return 0;
        if (G != G_) { snprintf(err_, sizeof err_, "row length changed"); return -1; }
        const size_t row = (size_t)G * sizeof(double);
        double *send_up = rowbuf_[0], *send_down = rowbuf_[1], *recv_top = rowbuf_[2], *recv_bot = rowbuf_[3];
        if (has_top_ && hip(hipMemcpyAsync(send_up, base, row, hipMemcpyDeviceToHost, s), "D2H first row")) return -1;
        if (has_bot_ && hip(hipMemcpyAsync(send_down, base + (rows - 1) * G, row, hipMemcpyDeviceToHost, s), "D2H last row")) return -1;
        if (hip(hipStreamSynchronize(s), "sync")) return -1;
        if (ex_(user_, has_top_ ? send_up : nullptr, has_bot_ ? send_down : nullptr, has_top_ ? recv_top : nullptr,
                has_bot_ ? recv_bot : nullptr, G) != 0) {
            snprintf(err_, sizeof err_, "host halo-exchange callback failed");
            return -1;
        }
        if (has_top_ && hip(hipMemcpyAsync(top_dst, recv_top, row, hipMemcpyHostToDevice, s), "H2D top row")) return -1;
        if (has_bot_ && hip(hipMemcpyAsync(bot_dst, recv_bot, row, hipMemcpyHostToDevice, s), "H2D bottom row")) return -1;
        return hip(hipStreamSynchronize(s), "sync");
    }

private:
    int hip(hipError_t e, const char *what)
    {
        if (e == hipSuccess) return 0;
        snprintf(err_, sizeof err_, "%s: %s", what, hipGetErrorString(e));
        return -1;
    }
    cg_exchange_fn ex_ = nullptr;
    cg_allgather_fn ag_ = nullptr;
    void *user_ = nullptr;
    long G_ = 0;
    double *rowbuf_[4] = {nullptr, nullptr, nullptr, nullptr};
    double *hgather_ = nullptr;
};

} // namespace cg
