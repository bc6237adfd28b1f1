// batch_hub.h -- many sequential estimators (USAC, ARRSAC) sharing every kernel launch.
// Included by ransac_5pt.hip inside namespace mlpl, before arrsac_impl.h / usac_impl.h.
//
// USAC and ARRSAC are sequential programs whose control flow runs on the host (usac_impl.h, arrsac_impl.h): a run alternates between
// host decisions and small dependent launch chains (a speculative batch of samples: solver -> roots -> check; a local optimisation; a
// degeneracy test), each followed by a wait.  One run leaves the chip idle -- its kernels are a few hundred waves -- and a batch of B
// problems run one after the other costs B times that latency.  Here B runs advance TOGETHER:
//   * every kernel of those paths has the form  body(const Args &, block x, block y)  with two entry points generated from it: the
//     single launch  k<<<grid>>>(Args)  and the merged launch  k_batch<<<(max grid x, max grid y, items)>>>(Args items[])  in which
//     blockIdx.z selects the item (a run's launch) and blocks beyond an item's own grid return at once;
//   * a run issues its launches through a Launcher.  Alone, the Launcher launches and waits as before.  In a batch every run is a host
//     thread (its sequential code is unchanged: a thread is the run's stack), its Launcher RECORDS the launches of a chain, and its
//     wait hands the list to the hub and blocks;
//   * the hub -- the calling thread -- waits until every live run is blocked, groups the runs by the kernel sequence of their lists
//     (runs at the same point of their control flow have the same sequence), issues per group and position ONE merged launch, waits for
//     the stream once and wakes everybody.  Host decisions of different runs execute in parallel on the host cores in between.
// Results do not depend on the grouping: a run's launches see exactly the arguments it recorded, in its order.
#pragma once



namespace {

constexpr int kHubArgBytes = 640;   // largest Args struct (USAC's check / local-optimisation arguments carry the five 3 x 3 normalisations)
constexpr int kHubMaxGroups = 4;    // groups of one round run on this many streams side by side

struct KHdr {  // first member of every Args: the item's own grid
    int gx, gy;
};

// kernel table: filled by MLPL_HUB_KERNEL below
typedef void (*HubSingleFn)(const void *args, hipStream_t s);
typedef void (*HubBatchFn)(const void *d_items, int count, int gx, int gy, hipStream_t s);
struct HubKernel {
    HubSingleFn single;
    HubBatchFn batch;
    int arg_bytes;
};
constexpr int kHubMaxKernels = 48;
inline HubKernel *hub_kernels() {
    static HubKernel tab[kHubMaxKernels];
    return tab;
}

template <class Args, void (*Body)(const Args &, int, int), int kThreads>
__global__ __launch_bounds__(kThreads) void hub_single_kernel(Args a) {
    Body(a, blockIdx.x, blockIdx.y);
}
template <class Args, void (*Body)(const Args &, int, int), int kThreads>
__global__ __launch_bounds__(kThreads) void hub_batch_kernel(const Args *__restrict__ items) {
    const Args &a = items[blockIdx.z];
    if ((int)blockIdx.x >= a.hdr.gx || (int)blockIdx.y >= a.hdr.gy) return;
    Body(a, blockIdx.x, blockIdx.y);
}
template <class Args, void (*Body)(const Args &, int, int), int kThreads>
struct HubReg {
    static void single(const void *args, hipStream_t s) {
        const Args &a = *static_cast<const Args *>(args);
        hipLaunchKernelGGL((hub_single_kernel<Args, Body, kThreads>), dim3(a.hdr.gx, a.hdr.gy), dim3(kThreads), 0, s, a);
    }
    static void batch(const void *d_items, int count, int gx, int gy, hipStream_t s) {
        hipLaunchKernelGGL((hub_batch_kernel<Args, Body, kThreads>), dim3(gx, gy, count), dim3(kThreads), 0, s, static_cast<const Args *>(d_items));
    }
    HubReg(int kid) {
        static_assert(sizeof(Args) <= kHubArgBytes, "Args struct larger than a launch record");
        static_assert(std::is_trivially_copyable<Args>::value, "Args must be plain data");
        hub_kernels()[kid] = HubKernel{&single, &batch, (int)sizeof(Args)};
    }
};
#define MLPL_HUB_KERNEL(KID, ARGS, BODY, THREADS) static HubReg<ARGS, BODY, THREADS> hub_reg_##KID(KID)

struct HubLaunch {
    int kid;
    alignas(16) unsigned char args[kHubArgBytes];
};

// The runs of a batch are host threads; creating 128 of them costs ~2.5 ms, a fifth of a batch: they are kept (blocked) between calls.
class HubThreads {
   public:
    // run job(k) for k in [0, n) on n threads at once; returns at once (wait() joins the round)
    void start(int n, std::function<void(int)> job) {
        std::unique_lock<std::mutex> lk(m_);
        while ((int)th_.size() < n) {
            const int k = (int)th_.size();
            th_.emplace_back([this, k] { loop(k); });
        }
        job_ = std::move(job);
        active_ = n, pending_ = n;
        ++gen_;
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
    }
    ~HubThreads() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            cv_.notify_all();
        }
        for (auto &t : th_) t.join();
    }

   private:
    void loop(int k) {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || (gen_ != seen && k < active_); });
            if (stop_) return;
            seen = gen_;
            lk.unlock();
            job_(k);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::function<void(int)> job_;
    unsigned long long gen_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};

struct HubStreams {  // what the hub keeps between calls (owned by the context): helper streams -- groups of one round run side by side -- and the run threads
    hipEvent_t ev[kHubMaxGroups] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t aux[kHubMaxGroups] = {nullptr, nullptr, nullptr, nullptr};
    HubThreads threads;
};
inline HubStreams *hub_resources(mlpl_ctx *ctx) {
    if (!ctx->hub_streams) ctx->hub_streams = new HubStreams();
    return static_cast<HubStreams *>(ctx->hub_streams);
}

class BatchHub;
struct HubRun {  // one run's side of the hub
    std::vector<HubLaunch> list;
    bool blocked = false, finished = false;
    unsigned long long served = 0;  // rounds this run's list was executed in
    int rc = 0;                     // result of the round (a launch error)
};

class BatchHub {
   public:
    BatchHub(mlpl_ctx *ctx, hipStream_t s, int runs) : ctx_(ctx), s_(s), runs_((size_t)runs) {}
    HubRun &run(int i) { return runs_[(size_t)i]; }
    hipStream_t stream() const { return s_; }

    // called by a run's thread: hand over the recorded launches, block until they have executed
    int wait(HubRun &r) {
        std::unique_lock<std::mutex> lk(m_);
        const unsigned long long target = r.served + 1;
        r.blocked = true;
        ++blocked_;
        cv_hub_.notify_one();
        cv_runs_.wait(lk, [&] { return r.served >= target; });
        return r.rc;
    }
    void finish(HubRun &r) {
        std::lock_guard<std::mutex> lk(m_);
        r.finished = true;
        ++finished_;
        cv_hub_.notify_one();
    }
    // called by the hub thread: serve rounds until every run has finished.  Returns the first error.
    int serve() {
        int first_rc = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            const auto t_wait = std::chrono::steady_clock::now();
            cv_hub_.wait(lk, [&] { return blocked_ + finished_ == (int)runs_.size(); });
            host_us_ += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_wait).count();
            if (blocked_ == 0) break;
            std::vector<HubRun *> todo;
            for (auto &r : runs_)
                if (r.blocked) todo.push_back(&r);
            lk.unlock();
            const auto t_exec = std::chrono::steady_clock::now();
            const int rc = execute(todo);
            device_us_ += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_exec).count();
            lk.lock();
            if (rc && !first_rc) first_rc = rc;
            for (HubRun *r : todo) {
                r->list.clear();
                r->rc = rc;
                r->blocked = false;
                ++r->served;
            }
            blocked_ = 0;
            ++rounds_;
            cv_runs_.notify_all();
        }
        return first_rc;
    }
    long long rounds() const { return rounds_; }
    long long merged_launches() const { return merged_; }
    long long host_us() const { return host_us_; }      // the hub waiting for the runs' host work (all runs blocked = a round can start)
    long long device_us() const { return device_us_; }  // building, issuing and waiting for the merged launches

   private:
    int execute(const std::vector<HubRun *> &todo) {
        // groups of runs with the same kernel sequence
        std::vector<std::vector<HubRun *>> groups;
        for (HubRun *r : todo) {
            if (r->list.empty()) continue;
            bool placed = false;
            for (auto &g : groups) {
                const HubRun *h = g[0];
                if (h->list.size() != r->list.size()) continue;
                bool same = true;
                for (size_t j = 0; j < r->list.size() && same; ++j) same = h->list[j].kid == r->list[j].kid;
                if (same) {
                    g.push_back(r);
                    placed = true;
                    break;
                }
            }
            if (!placed) groups.push_back(std::vector<HubRun *>(1, r));
        }
        if (groups.empty()) return MLPL_OK;
        // all items of the round in one pinned block, one copy to the device
        size_t bytes = 0;
        for (auto &g : groups)
            for (size_t j = 0; j < g[0]->list.size(); ++j) bytes += ((size_t)hub_kernels()[g[0]->list[j].kid].arg_bytes * g.size() + 255) & ~(size_t)255;
        if (bytes > ctx_->hub_items_cap) {  // the item tables live in the context: no allocation on the steady path
            MLPL_HIP_TRY(hipDeviceSynchronize());
            if (ctx_->hub_items_host) MLPL_HIP_TRY(hipHostFree(ctx_->hub_items_host));
            if (ctx_->hub_items_dev) MLPL_HIP_TRY(hipFree(ctx_->hub_items_dev));
            ctx_->hub_items_host = nullptr, ctx_->hub_items_dev = nullptr, ctx_->hub_items_cap = 0;
            const size_t want = bytes * 2 + (1u << 20);
            MLPL_HIP_TRY(hipHostMalloc(&ctx_->hub_items_host, want, hipHostMallocDefault));
            MLPL_HIP_TRY(hipMalloc(&ctx_->hub_items_dev, want));
            ctx_->hub_items_cap = want;
        }
        unsigned char *h_items_ = static_cast<unsigned char *>(ctx_->hub_items_host), *d_items_ = static_cast<unsigned char *>(ctx_->hub_items_dev);
        size_t off = 0;
        struct Plan {
            int kid, count, gx, gy, stream;
            size_t off;
        };
        std::vector<Plan> plan;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            auto &g = groups[gi];
            for (size_t j = 0; j < g[0]->list.size(); ++j) {
                const int kid = g[0]->list[j].kid, ab = hub_kernels()[kid].arg_bytes;
                int gx = 0, gy = 0;
                for (size_t k = 0; k < g.size(); ++k) {
                    const unsigned char *a = g[k]->list[j].args;
                    std::memcpy(h_items_ + off + k * (size_t)ab, a, (size_t)ab);
                    const KHdr *h = reinterpret_cast<const KHdr *>(a);
                    gx = std::max(gx, h->gx), gy = std::max(gy, h->gy);
                }
                plan.push_back(Plan{kid, (int)g.size(), gx, gy, (int)(gi % kHubMaxGroups), off});
                off += ((size_t)ab * g.size() + 255) & ~(size_t)255;
            }
        }
        MLPL_HIP_TRY(hipMemcpyAsync(d_items_, h_items_, off, hipMemcpyHostToDevice, s_));
        // group gi runs on stream gi % kHubMaxGroups (0 = the caller's); the helper streams start behind the item copy and are joined at the end
        const int used = (int)std::min<size_t>(groups.size(), kHubMaxGroups);
        HubStreams *hs = hub_resources(ctx_);
        if (used > 1) {
            if (!hs->ev[0]) {
                for (int i = 0; i < kHubMaxGroups; ++i) {
                    MLPL_HIP_TRY(hipEventCreateWithFlags(&hs->ev[i], hipEventDisableTiming));
                    if (i) MLPL_HIP_TRY(hipStreamCreateWithFlags(&hs->aux[i], hipStreamNonBlocking));
                }
            }
            MLPL_HIP_TRY(hipEventRecord(hs->ev[0], s_));
            for (int i = 1; i < used; ++i) MLPL_HIP_TRY(hipStreamWaitEvent(hs->aux[i], hs->ev[0], 0));
        }
        for (const Plan &p : plan) {
            hipStream_t st = p.stream == 0 ? s_ : hs->aux[p.stream];
            if (p.gx > 0 && p.gy > 0) hub_kernels()[p.kid].batch(d_items_ + p.off, p.count, p.gx, p.gy, st);
            ++merged_;
        }
        MLPL_HIP_TRY(hipGetLastError());
        for (int i = 1; i < used; ++i) {
            MLPL_HIP_TRY(hipEventRecord(hs->ev[i], hs->aux[i]));
            MLPL_HIP_TRY(hipStreamWaitEvent(s_, hs->ev[i], 0));
        }
        MLPL_HIP_TRY(hipStreamSynchronize(s_));
        return MLPL_OK;
    }

   private:
    mlpl_ctx *ctx_;
    hipStream_t s_;
    std::vector<HubRun> runs_;
    std::mutex m_;
    std::condition_variable cv_hub_, cv_runs_;
    int blocked_ = 0, finished_ = 0;
    long long rounds_ = 0, merged_ = 0, host_us_ = 0, device_us_ = 0;
};

// A run's view: launch now and wait on the stream (hub == nullptr), or record and wait on the hub.
struct Launcher {
    hipStream_t s = nullptr;
    BatchHub *hub = nullptr;
    HubRun *run = nullptr;
    template <class Args>
    void launch(int kid, const Args &a) {
        if (a.hdr.gx <= 0 || a.hdr.gy <= 0) return;
        if (!hub) {
            hub_kernels()[kid].single(&a, s);
            return;
        }
        run->list.emplace_back();
        HubLaunch &L = run->list.back();
        L.kid = kid;
        std::memcpy(L.args, &a, sizeof(Args));
    }
    int sync() {
        if (!hub) {
            MLPL_HIP_TRY(hipGetLastError());
            MLPL_HIP_TRY(hipStreamSynchronize(s));
            return MLPL_OK;
        }
        return hub->wait(*run);
    }
};

}  // namespace
