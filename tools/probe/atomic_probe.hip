// Diagnostic (not part of the product): what fp32 atomic accumulation of split-K partials costs on MI355X, depending on which
// workgroups (XCDs) hit which output lines.  256 or 512 workgroups each add a 128x128 fp32 tile (64 KiB) as 256-byte wave
// instructions into a C of `tiles` tiles.  mode 0: workgroup -> tile = id % tiles (every XCD touches every tile);
// mode 1: tile = XCD-owned (all partials of a tile come from one XCD); mode 2: per-XCD slab (8 x C), disjoint lines per XCD;
// mode 3: plain 16-byte stores of the same partials into a [splits][C] slab (no atomics).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void probe(float* C, int tiles, int mode, int wgs) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile; float* base = C;
    if (mode == 0) tile = id % tiles;
    else if (mode == 1) { const int per = tiles / 8; tile = xcd * per + slot % per; }     // tiles % 8 == 0
    else if (mode == 2) { tile = slot % tiles; base = C + (size_t)xcd * tiles * 16384; }
    else { tile = 0; base = C + (size_t)id * 16384; }
    float* t = base + (size_t)tile * 16384;
    const float v = (float)(id + 1);
    if (mode == 3) {
        for (int r = wave; r < 128; r += 4) {       // 128 rows x 512 B
            *reinterpret_cast<f32x4*>(t + r * 128 + (lane & 31) * 4) = f32x4{v, v, v, v};   // (both half-waves write the same 512 B: half-rate, fine for a bound)
        }
    } else {
        for (int r = wave; r < 128; r += 4) {
            atomicAdd(t + r * 128 + lane, v);
            atomicAdd(t + r * 128 + 64 + lane, v);
        }
    }
}
int main() {
    float* C; hipMalloc(&C, (size_t)512 * 65536 + (64 << 20));
    hipMemset(C, 0, (size_t)512 * 65536 + (64 << 20));
    const char* names[] = {"all XCDs -> all tiles", "tile owned by one XCD", "per-XCD slab", "plain stores to a slab"};
    for (int wgs = 256; wgs <= 512; wgs *= 2)
    for (int tiles = 8; tiles <= 64; tiles *= 2)
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, C, tiles, mode, wgs);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("wgs=%d tiles=%2d (C = %4.1f MB) partials %.0f MB  %-26s %.1f us\n", wgs, tiles, tiles * 65536 / 1e6, wgs * 65536 / 1e6, names[mode], best * 1e3);
    }
    return 0;
}
