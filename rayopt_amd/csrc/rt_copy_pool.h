/*
 * rt_copy_pool.h -- the host's share of the PCIe pipelines (rt_h2d,
 * rt_d2h_jobs): memcpy between pageable memory and the pinned staging
 * buffers on a few threads.
 *
 * One core copies ~30 GB/s, PCIe 5 x16 moves ~55: the staging copy is the
 * slower half of the pipeline unless it runs on several cores.  Until round
 * 6 every chunk STARTED its threads (std::thread per 20-32 MB chunk): the
 * per-chunk trace of round 5 shows 0.07-0.12 ms of "start" beside 0.3 ms of
 * copying, and eight threads slower than four (6.7 against 5.4 ms per 240 MB
 * row) -- the spawn, not the copy.  Now the workers live as long as the
 * process, parked on a generation counter: they spin for about a millisecond
 * after a job (the next chunk of a transfer arrives within 0.3-0.6 ms) and
 * sleep on a condition variable after that, so an idle engine costs nothing.
 *
 * One job at a time per process; a second caller (another context on another
 * host thread) copies on its own thread instead of waiting.  After a fork the
 * child starts its own workers.
 */
#ifndef RT_COPY_POOL_H
#define RT_COPY_POOL_H

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string.h>
#include <thread>
#include <unistd.h>

#define RT_POOL_MAX 16

struct rt_copy_pool {
    std::thread th[RT_POOL_MAX];
    int n = 0; /* workers (the caller is one more pair of hands) */
    pid_t pid = 0;
    std::mutex owner; /* held from start to finish of a job */
    std::mutex m;
    std::condition_variable cv;
    std::atomic<unsigned long long> gen{0};
    std::atomic<int> left{0};
    std::atomic<bool> stop{false};
    /* the job: worker w copies [ (w + first) * part, ... ) */
    char *dst = nullptr;
    const char *src = nullptr;
    size_t len = 0, part = 0;
    int first = 0;

    void run(int w, unsigned long long seen)
    {
        /* `seen`: the generation at the worker's birth -- one that joins a
         * pool which has worked before must not take the last job for a new
         * one (its buffers may be gone) */
        for (;;) {
            /* hot for ~1 ms after the last job, then asleep */
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (gen.load(std::memory_order_acquire) == seen &&
                   !stop.load(std::memory_order_relaxed)) {
                __builtin_ia32_pause();
                if (!(++spins & 0x3ff) &&
                    std::chrono::steady_clock::now() - t0 >
                        std::chrono::microseconds(1000)) {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] {
                        return gen.load(std::memory_order_acquire) != seen ||
                               stop.load(std::memory_order_relaxed);
                    });
                }
            }
            if (stop.load(std::memory_order_relaxed))
                return;
            seen = gen.load(std::memory_order_acquire);
            const size_t off = (size_t)(w + first) * part;
            if (off < len)
                memcpy(dst + off, src + off,
                       len - off < part ? len - off : part);
            left.fetch_sub(1, std::memory_order_acq_rel);
        }
    }

    void ensure(int workers)
    {
        if (pid == getpid() && n >= workers)
            return;
        if (pid != getpid()) { /* first use, or the child of a fork: the
                                  parent's threads do not exist here */
            for (int w = 0; w < n; ++w)
                new (&th[w]) std::thread(); /* (handles of threads that do
                                               not exist in this process:
                                               forgotten, not detached) */
            n = 0;
            pid = getpid();
            gen.store(0);
            left.store(0);
        }
        /* (no job is in flight: the caller holds `owner`) */
        const unsigned long long born = gen.load(std::memory_order_acquire);
        for (; n < workers && n < RT_POOL_MAX; ++n)
            th[n] = std::thread([this, w = n, born] { run(w, born); });
    }

    ~rt_copy_pool()
    {
        if (pid != getpid())
            return; /* (threads of another process: nothing to join) */
        {
            std::lock_guard<std::mutex> lk(m);
            stop.store(true);
        }
        cv.notify_all();
        for (int w = 0; w < n; ++w)
            if (th[w].joinable())
                th[w].join();
    }
};

static rt_copy_pool g_copy_pool;

/* a staging copy on `nt` threads: `start` hands the workers their parts and
 * leaves the first part to the caller's `finish` (unless `all`: the caller is
 * about to block in a DMA for as long as this copy takes, so the workers take
 * everything), so that the caller can do something else in between (rt_d2h:
 * issue the next copy kernel) */
struct rt_copy_team {
    void *dst;
    const void *src;
    size_t first; /* bytes the caller copies in finish */
    bool pooled;
};

static void rt_copy_start(rt_copy_team *team, void *dst, const void *src,
                          size_t len, int nt, bool all = false)
{
    team->dst = dst;
    team->src = src;
    team->first = len;
    team->pooled = false;
    if (nt <= 1 || len < ((size_t)4 << 20))
        return;
    rt_copy_pool &P = g_copy_pool;
    if (!P.owner.try_lock())
        return; /* another transfer has the workers: copy alone */
    const int workers = all ? nt : nt - 1;
    P.ensure(workers);
    const int have = P.n < workers ? P.n : workers;
    const int hands = all ? have : have + 1;
    if (have < 1) {
        P.owner.unlock();
        return;
    }
    const size_t part = (len / hands + 4095) & ~(size_t)4095;
    P.dst = (char *)dst;
    P.src = (const char *)src;
    P.len = len;
    P.part = part;
    P.first = all ? 0 : 1;
    team->first = all ? 0 : (part < len ? part : len);
    team->pooled = true;
    P.left.store(P.n, std::memory_order_release); /* every worker answers */
    {
        std::lock_guard<std::mutex> lk(P.m);
        P.gen.fetch_add(1, std::memory_order_acq_rel);
    }
    P.cv.notify_all();
}

static void rt_copy_finish(rt_copy_team *team)
{
    if (team->first)
        memcpy(team->dst, team->src, team->first);
    if (!team->pooled)
        return;
    rt_copy_pool &P = g_copy_pool;
    unsigned spins = 0;
    while (P.left.load(std::memory_order_acquire) > 0) {
        __builtin_ia32_pause();
        if (!(++spins & 0xffff))
            std::this_thread::yield();
    }
    team->pooled = false;
    P.owner.unlock();
}

static void rt_memcpy_mt(void *dst, const void *src, size_t len, int nt)
{
    rt_copy_team team;
    rt_copy_start(&team, dst, src, len, nt);
    rt_copy_finish(&team);
}

#endif /* RT_COPY_POOL_H */
