// emu_runtime.cpp -- the thread machinery behind tools/emu/fakehip (test infrastructure).
#include <hip/hip_runtime.h>
#include <memory>

thread_local EmuThread emu_t;
thread_local int emu_site = 0;
size_t emu_dyn_lds_bytes = 0;
__attribute__((aligned(64))) char emu_dyn_lds[160 * 1024];

namespace {
struct Arg { unsigned idx; emu_dim3 bid, bdim, gdim; EmuWave *wave; EmuBarrier *bb; const std::function<void()> *body; };
void *thread_main(void *p) {
    Arg *a = (Arg *)p;
    emu_t.tid = emu_dim3(a->idx % a->bdim.x, (a->idx / a->bdim.x) % a->bdim.y, a->idx / (a->bdim.x * a->bdim.y));
    emu_t.bid = a->bid; emu_t.bdim = a->bdim; emu_t.gdim = a->gdim;
    emu_t.wave = a->wave; emu_t.block_bar = a->bb; emu_t.lane = (int)(a->idx & 63);
    (*a->body)();
    return 0;
}
}  // namespace

// Threads that return early stop taking part: a wavefront whose lanes ALL return is gone (supported); a wavefront that loses SOME
// lanes before a later rendezvous would deadlock -- the kernels here keep such lanes alive with a predicate instead.  The same
// holds for __syncthreads after an early return of whole wavefronts: not supported (it is a bug on the GPU as well).
void emu_run_block(unsigned n_threads, emu_dim3 bid, emu_dim3 bdim, emu_dim3 gdim, const std::function<void()> &body) {
    const unsigned n_waves = (n_threads + 63) / 64;
    std::vector<EmuWave> waves(n_waves);
    for (unsigned w = 0; w < n_waves; ++w) {
        const unsigned lanes = w + 1 < n_waves || n_threads % 64 == 0 ? 64 : n_threads % 64;
        if (lanes != 64) { fprintf(stderr, "emu: block size %u is not a multiple of 64 (partial wavefronts are not modelled)\n", n_threads); abort(); }
        waves[w].bar.init((int)lanes);
        pthread_mutex_init(&waves[w].mu, 0);
        for (int r = 0; r < 4; ++r) waves[w].rowbar[r].init(16);
    }
    EmuBarrier bb; bb.init((int)n_threads);
    std::vector<pthread_t> th(n_threads); std::vector<Arg> args(n_threads);
    pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstacksize(&at, 1 << 20);
    for (unsigned i = 0; i < n_threads; ++i) {
        args[i] = Arg{ i, bid, bdim, gdim, &waves[i / 64], &bb, &body };
        if (pthread_create(&th[i], &at, thread_main, &args[i]) != 0) { perror("pthread_create"); abort(); }
    }
    for (unsigned i = 0; i < n_threads; ++i) pthread_join(th[i], 0);
    pthread_attr_destroy(&at);
}

#include <mutex>
std::mutex &emu_launch_mutex() { static std::mutex m; return m; }
