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
//   * a run issues its launches through a Launcher.  Alone, the Launcher launches and waits as before.  In a batch every run is a
//     fiber on one of a few worker threads (its sequential code is unchanged: the fiber is the run's stack), its Launcher RECORDS the
//     launches of a chain, and its wait hands the list to the hub and yields;
//   * the hub -- the calling thread, or a lane's thread: up to four cohorts of runs are served side by side, so that one cohort's host
//     turns run beside another's launches -- waits until every live run of its cohort is blocked, groups the runs by the kernel sequence of their lists
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

// ---- the runs of a batch: fibers on a few worker threads ---------------------------------------------------------------------------------
// A run is a sequential program with its own stack that blocks 6-12 times per problem on the hub.  As one kernel thread per run (round 4's
// first form) every such wait was a futex sleep and a wake-up of 128 threads at once: measured 60-90 us of CPU time per wait, i.e. more
// than half of the 0.66 ms of host CPU a USAC run cost, and with 512 runs alive the batch ran into the CPU quota of its container (16
// cores: calls of 17 ms stretched to 60-80 ms whenever the period's quota was spent).  So a run is a FIBER (ucontext: its own 1 MiB stack,
// mapped once and kept): worker thread w of a pool owns the runs w, w + W, ... and switches into whichever of them can go on; a run that
// waits for the hub switches back to its worker (two user-level context switches, no system call apart from glibc's signal-mask
// bookkeeping), and a worker sleeps -- on the hub's generation word, one futex -- only when all of its runs wait.  A run's code does not
// change: its stack is the fiber's.  A fiber stays on its worker, so thread-local state (the error text) behaves as before.
inline long futex_wait_u32(const std::atomic<uint32_t> *w, uint32_t v) {
    return syscall(SYS_futex, reinterpret_cast<const uint32_t *>(w), FUTEX_WAIT_PRIVATE, v, nullptr, nullptr, 0);
}
inline long futex_wake_u32(const std::atomic<uint32_t> *w, int n) {
    return syscall(SYS_futex, reinterpret_cast<const uint32_t *>(w), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}

constexpr size_t kFiberStackBytes = 1u << 20;
constexpr int kHubWorkersDefault = 16;  // worker threads of one pool (option hub_workers)

class HubThreads;
struct Fiber {
    enum State { kIdle, kRunnable, kRunning, kWaiting, kDone };
    ucontext_t uc;
    ucontext_t *back = nullptr;  // the owning worker's scheduler context
    char *stack = nullptr;
    State state = kIdle;
    int index = 0;
    const std::atomic<uint32_t> *word = nullptr;  // kWaiting: resumable once *word != value
    uint32_t value = 0;
    HubThreads *pool = nullptr;
};
inline Fiber *&hub_current_fiber() {
    static thread_local Fiber *f = nullptr;
    return f;
}
// Block the caller until *word != value: a fiber yields to its worker, a plain thread sleeps on the word.
inline void hub_block_until_changed(const std::atomic<uint32_t> *word, uint32_t value) {
    Fiber *f = hub_current_fiber();
    if (!f) {
        while (word->load(std::memory_order_acquire) == value) futex_wait_u32(word, value);
        return;
    }
    f->word = word, f->value = value, f->state = Fiber::kWaiting;
    swapcontext(&f->uc, f->back);
}

class HubThreads {
   public:
    // run job(k) for k in [0, n), every k as a fiber, on `workers` threads (0 = kHubWorkersDefault; at most n); returns at once (wait() joins)
    void start(int n, std::function<void(int)> job, int workers = 0) {
        std::unique_lock<std::mutex> lk(m_);
        const int W = std::max(1, std::min(n, workers > 0 ? workers : kHubWorkersDefault));
        while ((int)fibers_.size() < n) {
            std::unique_ptr<Fiber> f(new Fiber());
            void *st = mmap(nullptr, kFiberStackBytes + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK | MAP_NORESERVE, -1, 0);
            if (st == MAP_FAILED) throw std::bad_alloc();
            if (mprotect(st, 4096, PROT_NONE) != 0) {  // guard page below the stack
                munmap(st, kFiberStackBytes + 4096);
                throw std::bad_alloc();
            }
            f->stack = static_cast<char *>(st) + 4096;
            f->pool = this;
            fibers_.push_back(std::move(f));
        }
        while ((int)workers_.size() < W) {
            const int w = (int)workers_.size();
            workers_.emplace_back(new Worker());
            workers_.back()->th = std::thread([this, w] { loop(w); });
        }
        job_ = std::move(job);
        n_ = n, active_ = W, pending_ = W;
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
        for (auto &w : workers_) w->th.join();
        for (auto &f : fibers_) munmap(f->stack - 4096, kFiberStackBytes + 4096);
    }

   private:
    struct Worker {
        std::thread th;
        ucontext_t sched;
    };
    static void entry(unsigned hi, unsigned lo) {
        Fiber *f = reinterpret_cast<Fiber *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
        f->pool->job_(f->index);
        f->state = Fiber::kDone;
    }  // returns through uc_link into the worker's scheduler
    void serve_fibers(int w, int n, int W) {
        ucontext_t &sched = workers_[(size_t)w]->sched;
        for (int k = w; k < n; k += W) {
            Fiber &f = *fibers_[(size_t)k];
            getcontext(&f.uc);
            f.uc.uc_stack.ss_sp = f.stack, f.uc.uc_stack.ss_size = kFiberStackBytes, f.uc.uc_link = &sched;
            f.back = &sched, f.index = k, f.state = Fiber::kRunnable;
            const uintptr_t p = reinterpret_cast<uintptr_t>(&f);
            makecontext(&f.uc, reinterpret_cast<void (*)()>(&HubThreads::entry), 2, (unsigned)(p >> 32), (unsigned)(p & 0xffffffffu));
        }
        for (;;) {
            bool progressed = false, live = false, one_word = true;
            const Fiber *sleeper = nullptr;
            for (int k = w; k < n; k += W) {
                Fiber &f = *fibers_[(size_t)k];
                if (f.state == Fiber::kDone) continue;
                live = true;
                if (f.state == Fiber::kWaiting && f.word->load(std::memory_order_acquire) == f.value) {
                    if (!sleeper) sleeper = &f;
                    else if (sleeper->word != f.word) one_word = false;
                    continue;
                }
                f.state = Fiber::kRunning;
                hub_current_fiber() = &f;
                swapcontext(&sched, &f.uc);
                hub_current_fiber() = nullptr;
                progressed = true;
            }
            if (!live) return;
            if (!progressed && sleeper) {  // every live fiber waits (returns at once if the word has moved on)
                if (one_word)
                    futex_wait_u32(sleeper->word, sleeper->value);
                else {  // (fibers of one pool waiting on different words -- not what the hub does: poll them all)
                    const timespec ts{0, 50000};
                    syscall(SYS_futex, reinterpret_cast<const uint32_t *>(sleeper->word), FUTEX_WAIT_PRIVATE, sleeper->value, &ts, nullptr, 0);
                }
            }
        }
    }
    void loop(int w) {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || (gen_ != seen && w < active_); });
            if (stop_) return;
            seen = gen_;
            const int n = n_, W = active_;
            lk.unlock();
            serve_fibers(w, n, W);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::unique_ptr<Worker>> workers_;
    std::vector<std::unique_ptr<Fiber>> fibers_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::function<void(int)> job_;
    unsigned long long gen_ = 0;
    int n_ = 0, active_ = 0, pending_ = 0;
    bool stop_ = false;
};

constexpr int kHubLanes = 8;         // most cohorts of runs served side by side (option hub_lanes): while one cohort's launches execute, the others' runs do their host work
constexpr int kHubLanesDefault = 4;  // ... and how many without the option (USAC with REF_WEIGHTS: six, hub_usac_lanes_default)
inline int hub_usac_lanes_default(int refine) { return refine == 0 ? 6 : kHubLanesDefault; }

struct HubLane {  // what one cohort's hub keeps between calls (owned by the context)
    hipEvent_t ev[kHubMaxGroups] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t aux[kHubMaxGroups] = {nullptr, nullptr, nullptr, nullptr};  // groups of one round run side by side ([0] unused: the hub's own stream)
    hipStream_t own = nullptr;  // the stream of a lane > 0 (lane 0 serves on the caller's stream)
    hipEvent_t done = nullptr;  // end of a round (created with hipEventBlockingSync: the lane's thread sleeps instead of spinning, option hub_blocking_sync)
    void *items_host = nullptr, *items_dev = nullptr;  // item tables of the merged launches of a round
    size_t items_cap = 0;
    HubThreads threads;  // the cohort's run threads
};
struct HubStreams {
    HubLane lane[kHubLanes];
    hipStream_t copy = nullptr;  // device -> host copies that travel beside the lanes' work (the correspondences of a USAC batch)
    hipStream_t feed = nullptr;  // the producer side of a cohort feed (the matching of later cohorts beside the estimators of earlier ones)
    hipEvent_t feed_start = nullptr;
    HubThreads &threads = lane[0].threads;  // (helpers that only need a thread pool between rounds)
};
inline HubStreams *hub_resources(mlpl_ctx *ctx) {
    if (!ctx->hub_streams) ctx->hub_streams = new HubStreams();
    return static_cast<HubStreams *>(ctx->hub_streams);
}
// the stream lane `l` serves on: the caller's for lane 0, the lane's own otherwise (created on first use)
inline int hub_lane_stream(mlpl_ctx *ctx, int l, hipStream_t caller, hipStream_t *out) {
    if (l == 0) {
        *out = caller;
        return MLPL_OK;
    }
    HubLane &L = hub_resources(ctx)->lane[l];
    if (!L.own) MLPL_HIP_TRY(hipStreamCreateWithFlags(&L.own, hipStreamNonBlocking));
    *out = L.own;
    return MLPL_OK;
}

// How a batched sequential estimator cuts its problems into cohorts (runs that advance together) and lanes (cohorts in flight); shared
// by the estimators and by producers that feed them cohort by cohort (pair_batch_usac.h).
// lanes_default: the estimator's own choice without the option (measured, tools/c5_lanes_sweep.py, ms per 512 image pairs at 4 / 6 / 8 lanes: USAC with
// REF_WEIGHTS 15.8 / 14.7-15.2 / 15.0 uniform, 17.7-20 / 16.4 / 16.5 PROSAC -- its rounds are half host work, which more cohorts overlap;
// the usac5_* refinements 46.5 / 52-56 / 54 and ARRSAC 20.6 / 20.5 / 22.8 -- their rounds are chains of small launches that only get in each other's way)
inline int hub_cohort_size(const mlpl_ctx *ctx, int B, int cohort_default, int *n_cohorts_out, int *lanes_out, int lanes_default = kHubLanesDefault) {
    const int cohort_max = ctx->opt_hub_cohort > 0 ? ctx->opt_hub_cohort : cohort_default;
    const int lanes_wanted = ctx->opt_hub_lanes > 0 ? std::min(ctx->opt_hub_lanes, kHubLanes) : std::min(lanes_default, kHubLanes);
    const int cohort = B >= 8 * lanes_wanted ? std::min(cohort_max, (B + lanes_wanted - 1) / lanes_wanted) : B;
    const int n_cohorts = (B + cohort - 1) / cohort;
    if (n_cohorts_out) *n_cohorts_out = n_cohorts;
    if (lanes_out) *lanes_out = std::min(lanes_wanted, n_cohorts);
    return cohort;
}
// A producer that delivers the problems cohort by cohort (round 5: the image-pair entries match cohort c + 1 while the estimators of
// cohort c run).  ready[c]: an event behind everything cohort c's problems need on the device; on_ready(c, lane): called by the lane's
// thread once that event has completed, before the cohort's runs are built -- it fills the host-side inputs of problems
// [c * cohort, ...) (counts[b], parameters); its pool argument is the lane's idle run-thread pool.  The estimator then sizes its
// per-run blocks for `stride` correspondences, because the counts are not known up front.
struct CohortFeed {
    int cohort = 0, n_cohorts = 0;
    const hipEvent_t *ready = nullptr;
    std::function<int(int c, HubThreads &pool)> on_ready;
};

class BatchHub;
struct HubRun {  // one run's side of the hub
    std::vector<HubLaunch> list;
    bool blocked = false, finished = false;
    int rc = 0;  // result of the round (a launch error)
};

class BatchHub {
   public:
    BatchHub(mlpl_ctx *ctx, hipStream_t s, int runs, int lane = 0) : s_(s), lane_(&hub_resources(ctx)->lane[lane]), runs_((size_t)runs), pending_(runs), blocking_(ctx->opt_hub_blocking_sync != 0) {}
    HubRun &run(int i) { return runs_[(size_t)i]; }
    hipStream_t stream() const { return s_; }

    // called by a run: hand over the recorded launches, block until they have executed.  pending_ counts the runs that are neither
    // blocked nor finished; the run that brings it to zero wakes the hub; the hub's round ends by advancing gen_, which releases every
    // run that blocked before it (no lock: a run's list / flags are published by its decrement of pending_ and read by the hub after it
    // has seen zero).
    int wait(HubRun &r) {
        const uint32_t g = gen_.load(std::memory_order_acquire);
        r.blocked = true;
        if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) wake_hub();
        hub_block_until_changed(&gen_, g);
        return r.rc;
    }
    void finish(HubRun &r) {
        r.finished = true;
        if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) wake_hub();
    }
    // called by the hub thread: serve rounds until every run has finished.  Returns the first error.
    int serve() {
        int first_rc = 0;
        for (;;) {
            const auto t_wait = std::chrono::steady_clock::now();
            for (;;) {
                const uint32_t w = hub_word_.load(std::memory_order_acquire);
                if (pending_.load(std::memory_order_acquire) == 0) break;
                futex_wait_u32(&hub_word_, w);
            }
            host_us_ += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_wait).count();
            std::vector<HubRun *> todo;
            for (auto &r : runs_)
                if (r.blocked) todo.push_back(&r);
            if (todo.empty()) break;
            const auto t_exec = std::chrono::steady_clock::now();
            const int rc = execute(todo);
            device_us_ += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_exec).count();
            if (rc && !first_rc) first_rc = rc;
            for (HubRun *r : todo) {
                r->list.clear();
                r->rc = rc;
                r->blocked = false;
            }
            ++rounds_;
            pending_.store((int)todo.size(), std::memory_order_release);  // they are about to run again
            gen_.fetch_add(1, std::memory_order_release);
            futex_wake_u32(&gen_, INT_MAX);
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
        if (bytes > lane_->items_cap) {  // the item tables live in the context: no allocation on the steady path
            MLPL_HIP_TRY(hipStreamSynchronize(s_));
            if (lane_->items_host) MLPL_HIP_TRY(hipHostFree(lane_->items_host));
            if (lane_->items_dev) MLPL_HIP_TRY(hipFree(lane_->items_dev));
            lane_->items_host = nullptr, lane_->items_dev = nullptr, lane_->items_cap = 0;
            const size_t want = bytes * 2 + (1u << 20);
            MLPL_HIP_TRY(hipHostMalloc(&lane_->items_host, want, hipHostMallocDefault));
            MLPL_HIP_TRY(hipMalloc(&lane_->items_dev, want));
            lane_->items_cap = want;
        }
        unsigned char *h_items_ = static_cast<unsigned char *>(lane_->items_host), *d_items_ = static_cast<unsigned char *>(lane_->items_dev);
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
        HubLane *hs = lane_;
        if (used > 1) {
            for (int i = 0; i < kHubMaxGroups; ++i) {  // every handle on its own: a creation that failed once is tried again, never used as null
                if (!hs->ev[i]) MLPL_HIP_TRY(hipEventCreateWithFlags(&hs->ev[i], hipEventDisableTiming));
                if (i && !hs->aux[i]) MLPL_HIP_TRY(hipStreamCreateWithFlags(&hs->aux[i], hipStreamNonBlocking));
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
        if (blocking_) {
            if (!lane_->done) MLPL_HIP_TRY(hipEventCreateWithFlags(&lane_->done, hipEventBlockingSync | hipEventDisableTiming));
            MLPL_HIP_TRY(hipEventRecord(lane_->done, s_));
            MLPL_HIP_TRY(hipEventSynchronize(lane_->done));
        } else
            MLPL_HIP_TRY(hipStreamSynchronize(s_));
        return MLPL_OK;
    }

   private:
    hipStream_t s_;
    HubLane *lane_;
    std::vector<HubRun> runs_;
    void wake_hub() {
        hub_word_.fetch_add(1, std::memory_order_release);
        futex_wake_u32(&hub_word_, 1);
    }
    std::atomic<int> pending_;                      // runs neither blocked nor finished
    bool blocking_;
    std::atomic<uint32_t> gen_{0}, hub_word_{0};    // futex words: rounds served (the runs wait on it) / wake-ups of the hub
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
