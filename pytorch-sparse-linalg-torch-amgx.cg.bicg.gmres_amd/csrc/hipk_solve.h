// hipk_solve.h -- host-side helpers shared by the device-resident solve loops.
#pragma once
#include <math.h>
#include <stdlib.h>

#include <chrono>
#include <tuple>
#include <utility>
#include <vector>

#include "hipk_common.h"

struct hipk_event_pair {
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t create() {
        hipError_t e = hipEventCreate(&a);
        if (e != hipSuccess) return e;
        return hipEventCreate(&b);
    }
    ~hipk_event_pair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

// Kernel durations for bench.py (params.profile = the KIND of kernel to report: 1 SpMV, 2 CG update, 3 CG direction, 4 the
// scalars launch of the flat direction step).  The selected launches go through hipExtLaunchKernel with a start and a stop event
// bound to that dispatch; the figure is stop - start, RAW: nothing is calibrated or subtracted (until version 200 hipEventRecord
// pairs bracketed the launches and an estimated pair overhead of 3.4-3.8 us was subtracted: 0.876 printed where the profiler said
// 0.846).  What it measures (same box, N = 4 M CG loop, against `rocprofv3 --kernel-trace` averages of the same command):
//   only the selected kind timed (the default)   direction 25.2 / update 17.5 / coded SpMV 16.5 us   profiler 23.7 / 16.0 / 15.9
//   every launch of the iteration timed (chain)  23.2 / 15.5 / 14.7 us, and the iteration stretches from 56.1 to 69.2 us
// i.e. the start stamp is taken when the dispatch is picked up, up to ~1.5 us before its first wave runs when the previous kernel
// is still draining; with events on every dispatch the kernels no longer overlap their neighbours' tails (shorter spans, 4-5 us of
// idle between them).  The raw single-kind figure is therefore CONSERVATIVE by 0.6-1.5 us (4-9 % on these 16-25 us kernels, < 1 %
// on the N = 64 M ones); bench.py prints it and, beside it, the rocprofv3 average of the committed profile of the same build.
// `chain` (all launches timed; stats.spmv_ms_avg = stop-to-stop) is kept for such experiments only.
struct hipk_spmv_profiler {
    static constexpr int kMax = 256;        // launches of the selected kind
    static constexpr int kSlots = 5 * kMax; // all timed launches (chain mode: a CG iteration has 3-4)
    bool on;
    bool chain = false;
    int select;
    std::vector<hipEvent_t> ev;
    std::vector<unsigned char> kind;
    int used = 0, used_sel = 0;
    explicit hipk_spmv_profiler(int profile, bool chain_mode = false) : on(profile != 0), chain(chain_mode), select(profile) {
        if (!on) return;
        ev.resize(2 * (chain ? kSlots : kMax), nullptr);
        kind.resize(chain ? kSlots : kMax, 0);
        for (auto &e : ev)
            if (hipEventCreate(&e) != hipSuccess) {
                on = false;
                break;
            }
    }
    ~hipk_spmv_profiler() {
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    // the event pair of the next launch if it is to be timed (kind 0: a helper launch, timed only as a link of the chain)
    bool slot(int k, hipEvent_t *e0, hipEvent_t *e1) {
        if (!on || used_sel >= kMax || (size_t)used >= kind.size()) return false;
        if (!chain && k != select) return false;
        *e0 = ev[2 * used];
        *e1 = ev[2 * used + 1];
        kind[used] = (unsigned char)k;
        ++used;
        if (k == select) ++used_sel;
        return true;
    }
    // valid: number of leading launches of the selected kind that did real work
    hipError_t collect(hipk_stats *st, int64_t valid = INT64_MAX) {
        st->spmv_ms_avg = 0.0;
        st->spmv_profiled = 0;
        st->dispatch_span_ms_avg = 0.0;
        if (!on) return hipSuccess;
        double span = 0.0, occ = 0.0;
        int cnt = 0, nocc = 0;
        for (int i = 0; i < used && cnt < valid; ++i) {
            if (kind[i] != select) continue;
            float ms = 0.f;
            hipError_t e = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
            if (e != hipSuccess) return e;
            span += ms;
            ++cnt;
            if (chain && i > 0) {
                e = hipEventElapsedTime(&ms, ev[2 * (i - 1) + 1], ev[2 * i + 1]);
                if (e != hipSuccess) return e;
                occ += ms;
                ++nocc;
            }
        }
        if (cnt > 0) st->dispatch_span_ms_avg = span / cnt;
        st->spmv_ms_avg = (chain && nocc > 0) ? occ / nocc : st->dispatch_span_ms_avg;
        st->spmv_profiled = cnt;
        return hipSuccess;
    }
};

enum { HIPK_K_AUX = 0, HIPK_K_SPMV = 1, HIPK_K_UPDATE = 2, HIPK_K_DIRECTION = 3, HIPK_K_SCALARS = 4 };

#ifdef __HIPCC__
// kern<<<grid, block, shm, s>>>(args...), timed by `prof` (may be null) when it wants launches of this kind
template <typename... KA, size_t... I>
static inline void hipk_launch_timed_impl(hipk_spmv_profiler *prof, int kind, void (*kern)(KA...), dim3 grid, dim3 block, size_t shm,
                                          hipStream_t s, std::tuple<KA...> &t, std::index_sequence<I...>) {
    void *argv[sizeof...(KA) > 0 ? sizeof...(KA) : 1] = {(void *)&std::get<I>(t)...};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof && prof->slot(kind, &e0, &e1))
        (void)hipExtLaunchKernel((const void *)kern, grid, block, argv, shm, s, e0, e1, 0);
    else
        (void)hipLaunchKernel((const void *)kern, grid, block, argv, shm, s);
}
template <typename... KA, typename... A>
static inline void hipk_launch_timed(hipk_spmv_profiler *prof, int kind, void (*kern)(KA...), dim3 grid, dim3 block, size_t shm,
                                     hipStream_t s, A... args) {
    std::tuple<KA...> t(static_cast<KA>(args)...);
    hipk_launch_timed_impl(prof, kind, kern, grid, block, shm, s, t, std::index_sequence_for<KA...>{});
}
#endif

// Asynchronous reads of one device word (the stop word) into pinned host memory, at most
// two in flight: the host learns "the loop has stopped" one batch late and never stalls
// the GPU queue.
struct hipk_poller {
    int64_t *host;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int head = 0, count = 0;
    explicit hipk_poller(int64_t *pinned) : host(pinned) {}
    hipError_t create() {
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    ~hipk_poller() {
        for (int i = 0; i < 2; ++i)
            if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    hipError_t post(const int64_t *dev_word, int64_t, hipStream_t s) {
        const int slot = (head + count) & 1;
        hipError_t e = hipMemcpyAsync(&host[slot], dev_word, sizeof(int64_t), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
        e = hipEventRecord(ev[slot], s);
        if (e != hipSuccess) return e;
        ++count;
        return hipSuccess;
    }
    void harvest(int64_t *stop) {
        const int64_t v = host[head];
        if (v < *stop) *stop = v;
        head = (head + 1) & 1;
        --count;
    }
    hipError_t wait_oldest_if_full(int64_t *stop) {
        while (count > 0) {
            hipError_t q = hipEventQuery(ev[head]);
            if (q == hipSuccess) {
                harvest(stop);
            } else if (q == hipErrorNotReady) {
                break;
            } else {
                return q;
            }
        }
        if (count == 2) {
            hipError_t e = hipEventSynchronize(ev[head]);
            if (e != hipSuccess) return e;
            harvest(stop);
        }
        return hipSuccess;
    }
    hipError_t drain(int64_t *stop) {
        while (count > 0) {
            hipError_t e = hipEventSynchronize(ev[head]);
            if (e != hipSuccess) return e;
            harvest(stop);
        }
        return hipSuccess;
    }
};

// ---- pacing of a device-resident loop -------------------------------------------------------------------------
// The loop's deciding kernel (one thread of it) reports to a word of PINNED HOST memory: the number of completed
// iterations, or HIPK_SIG_STOP | stop_it once the stop rule has fired (hipk_signal).  The host reads that word
// before enqueueing an iteration -- a plain load, no stream operation -- and (a) keeps at most `window` iterations
// queued ahead of the GPU, (b) stops enqueueing as soon as the loop has stopped.  A short solve (15 iterations of a
// 100-row system) therefore launches 15 + window iterations instead of two polling batches of 64: 1.4 -> 0.4 ms.
// If the word does not move for 200 ms (another stream hogging the GPU, or a platform on which device stores to
// pinned memory are not visible before the kernel ends) the pacer falls back to the stream-ordered poller above.
#define HIPK_SIG_STOP ((int64_t)1 << 62)

#ifdef __HIPCC__
__device__ __forceinline__ void hipk_signal(int64_t *sig, int64_t v) {
    if (sig) __hip_atomic_store(sig, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

struct hipk_pacer {
    hipk_poller poll;
    int64_t *sig;             // pinned host word (device-visible at the same address), or null
    const int64_t *dev_stop;  // device stop word, for the fallback
    int64_t check, window, next_post = 0;
    int64_t timeout_us = 200000;  // HIPK_PACE_TIMEOUT_US: how long the word may stand still before the fallback (tests: 0)
    bool live;
    // pinned: the handle's 16-word block; words 0-1 belong to the poller, word 2 is the signal
    hipk_pacer(int64_t *pinned, const int64_t *dev_stop_word, int64_t check_every, int64_t win = 8)
        : poll(pinned), sig(pinned + 2), dev_stop(dev_stop_word), check(check_every), window(win) {
        const char *e = getenv("HIPK_HOST_SIGNAL");
        live = !(e && e[0] == '0');
        if (!live) sig = nullptr;
        if (const char *w = getenv("HIPK_PACE_TIMEOUT_US")) {
            const long v = atol(w);
            if (v >= 0) timeout_us = v;
        }
        if (const char *w = getenv("HIPK_PACE_WINDOW")) {
            const long v = atol(w);
            if (v >= 1 && v <= 4096) window = v;
        }
    }
    hipError_t create() {
        if (sig) __atomic_store_n(sig, (int64_t)0, __ATOMIC_RELEASE);  // before the start kernel is enqueued
        return poll.create();
    }
    int64_t *device_sig() const { return sig; }
    // Call before enqueueing iteration `it`; *stop <= it afterwards means: do not enqueue it.
    hipError_t gate(int64_t it, hipStream_t s, int64_t *stop) {
        if (live) {
            int64_t v = __atomic_load_n(sig, __ATOMIC_ACQUIRE);
            if (!(v & HIPK_SIG_STOP) && v < it - window) {
                int64_t last = v;
                auto t0 = std::chrono::steady_clock::now();
                for (unsigned spins = 1;; ++spins) {
                    v = __atomic_load_n(sig, __ATOMIC_ACQUIRE);
                    if ((v & HIPK_SIG_STOP) || v >= it - window) break;
                    if ((spins & 1023u) == 0) {
                        const auto now = std::chrono::steady_clock::now();
                        if (v != last) {
                            last = v;
                            t0 = now;
                        } else if (now - t0 >= std::chrono::microseconds(timeout_us)) {
                            live = false;
                            next_post = it;
                            break;
                        }
                    }
                }
            }
            if (v & HIPK_SIG_STOP) {
                const int64_t at = v & ~HIPK_SIG_STOP;
                if (at < *stop) *stop = at;
                return hipSuccess;
            }
            if (live) return hipSuccess;
        }
        if (it >= next_post) {  // stream-ordered polling, two reads in flight
            hipError_t e = poll.post(dev_stop, it, s);
            if (e != hipSuccess) return e;
            next_post = it + check;
            return poll.wait_oldest_if_full(stop);
        }
        return hipSuccess;
    }
};

static inline size_t hipk_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// torch.maximum semantics (NaN wins).
static inline double hipk_tmax(double a, double b) {
    if (isnan(a) || isnan(b)) return NAN;
    return a > b ? a : b;
}
// sqrt(torch.clamp(v, min=0)) as `_norm` does (TSL:154-162); NaN stays NaN.
static inline double hipk_norm_from_sq(double v) { return sqrt(v < 0.0 ? 0.0 : v); }

// `_isolve` epilogue (TSL:1007-1016): info from the TRUE residual.
static inline void hipk_finish_isolve_stats(hipk_stats *st, const hipk_params *prm, double bs, double res2,
                                            double xx, int64_t iterations, int64_t matvecs) {
    st->iterations = iterations;
    st->matvecs = matvecs;
    st->b_norm = hipk_norm_from_sq(bs);
    st->residual_norm = hipk_norm_from_sq(res2);
    st->x_norm = hipk_norm_from_sq(xx);
    // torch.tensor(tol) is an fp32 tensor (TSL:1010-1011)
    st->threshold = hipk_tmax((double)(float)prm->tol * st->b_norm, (double)(float)prm->atol);
    const bool failed = isnan(st->x_norm) || (st->residual_norm > st->threshold);
    st->info = failed ? -1 : 0;
}
