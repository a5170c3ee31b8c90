// bf16 MFMA-fragment images of W_hh shared by the launch-per-step (lstm.hip) and persistent (lstm_persist.hip) recurrences.
#pragma once
#include "common.h"

namespace {

// side job of a fragment-image kernel: the launcher of a persistent recurrence needs its hand-off buffer preset (tags 0 / sentinels
// 0xFFFFFFFF) and its census counters zeroed before the launch -- done by the conversion kernel that runs in front of it anyway
// instead of two memset dispatches per launch (~4.6 us each on a stream whose every dispatch is serialised)
struct WfragAux { uint4* fill; unsigned long n16; unsigned value; unsigned* census; };
__device__ __forceinline__ void wfrag_aux(const WfragAux& a) {
    if (a.fill) {
        const uint4 v = {a.value, a.value, a.value, a.value};
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n16; i += (size_t)gridDim.x * blockDim.x) a.fill[i] = v;
    }
    if (a.census && blockIdx.x == 0 && threadIdx.x < 64) a.census[threadIdx.x] = 0u;          // 256 bytes of counters
}

// W_hh [4H][H] fp32 -> forward fragment image [H/4][H/32][64][8] bf16:
//   block jb, chunk c, lane (kg,li), e  <-  W_hh[(li>>2)*H + jb*4 + (li&3)][c*32 + kg*8 + e]
__global__ void make_wfrag_fwd(const float* __restrict__ w, unsigned short* __restrict__ out, int H, WfragAux aux) {
    wfrag_aux(aux);
    const size_t total = (size_t)4 * H * H;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t rest = i >> 9;
        const int nchunk = H >> 5;
        const int c = (int)(rest % nchunk), jb = (int)(rest / nchunk);
        const int li = lane & 15, kg = lane >> 4;
        const size_t row = (size_t)(li >> 2) * H + jb * 4 + (li & 3);
        out[i] = f2op16(w[row * H + c * 32 + kg * 8 + e]);
    }
}
// the same with the four GATES of a unit adjacent in the tile (column n = li: unit li >> 2, gate li & 3) -- lstm_persist_fwd_k: the
// epilogue then fetches the four gate partials of its element with ONE 16-byte LDS read per wave partial instead of four reads
//   block jb, chunk c, lane (kg,li), e  <-  W_hh[(li&3)*H + jb*4 + (li>>2)][c*32 + kg*8 + e]
__global__ void make_wfrag_fwd_ug(const float* __restrict__ w, unsigned short* __restrict__ out, int H, WfragAux aux) {
    wfrag_aux(aux);
    const size_t total = (size_t)4 * H * H;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t rest = i >> 9;
        const int nchunk = H >> 5;
        const int c = (int)(rest % nchunk), jb = (int)(rest / nchunk);
        const int li = lane & 15, kg = lane >> 4;
        const size_t row = (size_t)(li & 3) * H + jb * 4 + (li >> 2);
        out[i] = f2op16(w[row * H + c * 32 + kg * 8 + e]);
    }
}
// W_hh [4H][H] fp32 -> backward fragment image [H/16][4H/32][64][8] bf16:
//   tile jt, chunk c (over r = 0..4H), lane (kg,li), e  <-  W_hh[c*32 + kg*8 + e][jt*16 + li]
__global__ void make_wfrag_bwd(const float* __restrict__ w, unsigned short* __restrict__ out, int H, WfragAux aux) {
    wfrag_aux(aux);
    const size_t total = (size_t)4 * H * H;
    const int nchunk = (4 * H) >> 5;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t rest = i >> 9;
        const int c = (int)(rest % nchunk), jt = (int)(rest / nchunk);
        const int li = lane & 15, kg = lane >> 4;
        const size_t r = (size_t)c * 32 + kg * 8 + e;
        out[i] = f2op16(w[r * H + jt * 16 + li]);
    }
}

// W_hh [4H][H] fp32 -> reduce-scatter backward fragment image (lstm_persist_bwd_rs_k) [H/32][H/16][4][64][8] bf16:
//   CU slot q (units q*32 .. q*32+31), column tile jt, chunk g (= gate), lane (kg,li), e  <-  W_hh[g*H + q*32 + kg*8 + e][jt*16 + li]
// i.e. the B operand of  partial[row][j] = sum_c dgates[row][c] W_hh[grow(c)][j]  over the CU's OWN 128 gate rows c = g*32 + unit.
__global__ void make_wfrag_rs(const float* __restrict__ w, unsigned short* __restrict__ out, int H, WfragAux aux) {
    wfrag_aux(aux);
    const size_t total = (size_t)4 * H * H;
    const int ntile = H >> 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t rest = i >> 9;
        const int g = (int)(rest & 3), jt = (int)((rest >> 2) % ntile), q = (int)((rest >> 2) / ntile);
        const int li = lane & 15, kg = lane >> 4;
        const size_t r = (size_t)g * H + q * 32 + kg * 8 + e;
        out[i] = f2op16(w[r * H + jt * 16 + li]);
    }
}


}  // namespace
