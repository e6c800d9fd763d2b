// lane_emu.h -- test infrastructure: run per-lane device code on the host, one OS thread per lane of a 16-lane row, with a
// barrier at every cross-lane primitive.  The threads of a row execute the same (row-uniform) control flow, exactly what the
// device code relies on; each primitive publishes the lane's value, waits, reads its neighbour's, waits again.  Slow (tens of
// microseconds per primitive) but it executes the very source the GPU compiles -- indexing, list handling, reductions -- so
// that only the meaning of the hardware primitives themselves is left to check on the GPU.
#pragma once
#include <pthread.h>
#include <stdint.h>

#define BM2_EMU 1
#define BM2_DEV inline
#ifndef __restrict__
#define __restrict__
#endif

struct RowEmu {
    pthread_barrier_t bar;
    int slot[16];
};
static thread_local RowEmu *emu_row = nullptr;
static thread_local int emu_lane = 0;

static inline void emu_sync() { pthread_barrier_wait(&emu_row->bar); }

static inline int row_shr1(int v) {
    emu_row->slot[emu_lane] = v; emu_sync();
    const int r = emu_lane ? emu_row->slot[emu_lane - 1] : 0; emu_sync();
    return r;
}
static inline bool row_any(bool p) {
    emu_row->slot[emu_lane] = p; emu_sync();
    int any = 0;
    for (int i = 0; i < 16; ++i) any |= emu_row->slot[i];
    emu_sync();
    return any != 0;
}
static inline int row_xor(int v, int m) {
    emu_row->slot[emu_lane] = v; emu_sync();
    const int r = emu_row->slot[emu_lane ^ m]; emu_sync();
    return r;
}
static inline int row_first(int v) {
    emu_row->slot[emu_lane] = v; emu_sync();
    const int r = emu_row->slot[0]; emu_sync();
    return r;
}
