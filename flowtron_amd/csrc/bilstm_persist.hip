// Persistent bidirectional LSTM for the text encoder (flowtron.py:488, :505-512: nn.LSTM(512, 256, bidirectional=True)):
// ONE launch per pass for both directions and all time steps, instead of one launch per step (lstm.hip lstm_*_pair, 4.4 / 4.9 us
// per step at the launch floor).
//
// H = 256 is small enough for a much simpler organisation than lstm_persist.hip: there is NO workgroup-level step at all.
//   group      = (XCD x, direction d): batch rows [4x, 4x + 4) of that direction, for the whole sequence; formed at run time from
//                the workgroups' XCC ids (census, as in lstm_persist.hip), hand-off through that XCD's own L2;
//   forward    every WAVE is an independent agent that owns 4 hidden units (one 16-column MFMA tile = 4 units x 4 gates) with
//                the whole K = 256: 8 MFMAs per step, no K split, hence no LDS reduction and no barrier -- the four gates of a
//                unit are gathered into one lane with three DPP row shifts.  64 waves = 16 workgroups per group;
//   backward   a wave owns 16 hidden units (one column tile of W_hh^T, K = 4H = 1024: 32 MFMAs per step); 16 waves = 4 workgroups
//                per group;
//   hand-off   h_t / dgates_s as 8-byte {epoch, 16-bit pair} granules in the packed layout of lstm_persist.hip (one 16-byte load
//                per lane fetches four k-chunks of the group's four rows; DPP row shifts move them into the MFMA row positions),
//                plain stores, nt loads, two parity buffers, bounded spins with a status word;
//   HBM rows   (gx; saved gates, cell, dy) are fetched one step ahead into registers right after a step's poll has completed, so
//                that they have a whole step of slack before the next poll queues up behind them (vmcnt retires in order).
// The time index is uniform per group: forward direction t = s, reverse direction t = tg - 1 - s (tg = the longest of the group's
// four rows), a row takes part while t < len -- a row of the reverse direction simply starts late, with zero state.
// y / dgx rows of padded frames are zeroed by the launcher (memset), the kernels write valid frames only.
// Not bit-identical to the launch-per-step pair kernels (those split K over waves); same operand rounding, fp32 accumulation.
#include "common.h"
#include "lstm_images.h"

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BH = 256;                  // hidden size this file is built for
constexpr int NWG = 256;                 // workgroups launched: one per CU, 32 per XCD
constexpr int GRAN_F = 4 * BH / 2;       // granules of one forward state vector (4 rows x 256): 512
constexpr int GRAN_B = 4 * 4 * BH / 2;   // granules of one dgates vector (4 rows x 1024): 2048

template <int N>
__device__ __forceinline__ unsigned row_shl(unsigned v) {          // lane l <- lane l + N of the same 16-lane row
    if constexpr (N == 0) return v;
    else return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ float row_shl_f(float v) { return __uint_as_float(row_shl<N>(__float_as_uint(v))); }
template <int S>
__device__ __forceinline__ u32x4 shl4(u32x4 v) {
    return (u32x4){row_shl<S>(v[0]), row_shl<S>(v[1]), row_shl<S>(v[2]), row_shl<S>(v[3])};
}
// member j (0..3) of a load group = k-chunk 4 lg + j: its four rows moved to lanes li < 4 of every 16-lane row
__device__ __forceinline__ u32x4 member(u32x4 v, int j) {
    switch (j) {
        case 0: return v;
        case 1: return shl4<4>(v);
        case 2: return shl4<8>(v);
        default: return shl4<12>(v);
    }
}
// granule index of the operand pair (k, k + 1), k even, of row b (0..3): chunk c = k / 32 -> load group lg = c / 4, member j = c % 4,
// lane = kg 16 + j 4 + b; a lane's two 16-byte loads (halves) of a load group hold its eight k-values {v, tag, v, tag} x 2
__device__ __forceinline__ int gran_index(int b, int k) {
    const int c = k >> 5, kg = (k >> 3) & 3, e = k & 7;
    const int lg = c >> 2, j = c & 3, lane = kg * 16 + j * 4 + b;
    return ((((lg * 2 + (e >> 2)) * 64 + lane)) << 1) + ((e >> 1) & 1);
}

struct BiFwdP {
    const float* gx[2]; const int* lens; float* y; long ldy; float* gates[2]; float* cell[2];
    const unsigned short* wfrag[2];      // make_wfrag_fwd images of W_hh (forward, reverse)
    unsigned long long* gran;            // [2 parity][8 xcd][2 dir][GRAN_F]
    int* status; unsigned* census; int T, B; long timeout_ticks;
};
struct BiBwdP {
    const float* dy; long ldy; const int* lens; const float* gates[2]; const float* cell[2]; float* dgx[2];
    const unsigned short* wTfrag[2];     // make_wfrag_bwd images
    unsigned long long* gran;            // [2 parity][8 xcd][2 dir][GRAN_B]
    int* status; unsigned* census; int T, B; long timeout_ticks;
};

// XCD census: group = this workgroup's XCC id, slot = arrival order inside that XCD (lstm_persist.hip)
__device__ __forceinline__ bool join(unsigned* census, int* status, int tid, int& xcd, int& q) {
    __shared__ int slot[2];
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        slot[0] = (int)(xcc & 7u);
        slot[1] = (int)__hip_atomic_fetch_add(census + (xcc & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    xcd = __builtin_amdgcn_readfirstlane(slot[0]);
    q = __builtin_amdgcn_readfirstlane(slot[1]);
    if (q >= 32) {
        if (tid == 0) atomicExch(status, 2);
        return false;
    }
    return true;
}

// poll NL load groups (two 16-byte loads each) of this group's vector until every tag shows `epoch`; false = timed out
template <int NL>
__device__ __forceinline__ bool poll(u32x4 (&ld)[NL][2], __amdgpu_buffer_rsrc_t rs, int voff, unsigned epoch, long t_start,
                                     long timeout_ticks, int* status) {
    auto issue = [&](int g) {
#pragma unroll
        for (int h = 0; h < 2; ++h) ld[g][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (g * 2 + h) * 1024, 2);   // nt
    };
#pragma unroll
    for (int g = 0; g < NL; ++g) issue(g);
    unsigned ready = 0;
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int g = 0; g < NL; ++g) {
            if (!((ready >> g) & 1u)) {
                const bool ok = (ld[g][0][1] == epoch) & (ld[g][0][3] == epoch) & (ld[g][1][1] == epoch) & (ld[g][1][3] == epoch);
                if (__all(ok)) ready |= 1u << g;
            }
        }
        if (ready == (1u << NL) - 1u) return true;
        if ((spins & 15) == 15) {
            if (wall_clock64() - t_start > timeout_ticks || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                return false;
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g = 0; g < NL; ++g) {
            if (!((ready >> g) & 1u)) issue(g);
        }
    }
}

__global__ __launch_bounds__(256) void bilstm_persist_fwd_k(BiFwdP p) {
    extern __shared__ float occupancy_pad[];             // sized by the launcher so that one workgroup fills a CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int xcd, q;
    if (!join(p.census, p.status, tid, xcd, q)) return;
    const int dir = q >> 4, m = q & 15;                  // 16 workgroups per (XCD, direction)
    const int B = p.B, T = p.T, b0 = xcd * 4;
    const int jb = m * 4 + wave;                         // this wave's tile: units jb*4 .. jb*4 + 3
    int len[4], tg = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        len[r] = (b0 + r < B) ? min(p.lens[b0 + r], T) : 0;
        tg = len[r] > tg ? len[r] : tg;
    }
    bf16x8 w[8];
    {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.wfrag[dir]);
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c] = wf[((size_t)jb * 8 + c) * 64 + lane];
    }
    // The MFMA leaves row r of the tile in register r of lanes 0..15 (lane li <-> gate li / 4, unit li % 4).  One shuffle per row
    // hands row r to the 16-lane group kg = r, so that every lane group finishes ONE row: one gx load, one cell update and one
    // store of each kind per lane and step instead of four (the vector-memory instruction count is what a step costs here).
    const int mylen = kg == 0 ? len[0] : kg == 1 ? len[1] : kg == 2 ? len[2] : len[3];
    const int gcol = (li >> 2) * BH + jb * 4 + (li & 3);                 // column of gx of this lane: gate li / 4, unit li % 4
    const float* gxp = p.gx[dir];
    auto load_gx = [&](int t) { return (t >= 0 && t < mylen) ? gxp[((size_t)t * B + b0 + kg) * 4 * BH + gcol] : 0.f; };
    unsigned long long* gbase = p.gran + ((size_t)xcd * 2 + dir) * GRAN_F;          // + parity * 16 * GRAN_F
    __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) rs[par] = __builtin_amdgcn_make_buffer_rsrc(gbase + (size_t)par * 16 * GRAN_F, 0, GRAN_F * 8, 0x00020000);
    const int voff = lane * 16;
    const long t_start = wall_clock64();
    float c_state = 0.f, h_state = 0.f;
    float gxv = load_gx(dir ? tg - 1 : 0), gxn;
    const int u = jb * 4 + (li & 3);                     // (lanes li < 4 of every group own unit li)
    bool dead = false;
    for (int s = 0; s < tg; ++s) {
        const int t = dir ? tg - 1 - s : s;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        u32x4 ld[2][2];
        if (s > 0) {
            if (!poll<2>(ld, rs[(s - 1) & 1], voff, (unsigned)s, t_start, p.timeout_ticks, p.status)) { dead = true; break; }
        }
        gxn = load_gx(dir ? t - 1 : t + 1);              // next step's row: a whole step ahead of the poll that queues behind it
        if (s > 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const u32x4 pay = (u32x4){ld[c >> 2][0][0], ld[c >> 2][0][2], ld[c >> 2][1][0], ld[c >> 2][1][2]};
                acc = mfma16(__builtin_bit_cast(bf16x8, member(pay, c & 3)), w[c], acc);
            }
        }
        const float x1 = __shfl(acc[1], li, 64), x2 = __shfl(acc[2], li, 64), x3 = __shfl(acc[3], li, 64);
        const float pre_i = (kg == 0 ? acc[0] : kg == 1 ? x1 : kg == 2 ? x2 : x3) + gxv;     // (row kg, gate li / 4, unit li % 4)
        float pre[4] = {pre_i, row_shl_f<4>(pre_i), row_shl_f<8>(pre_i), row_shl_f<12>(pre_i)};   // lanes li < 4: the four gates of unit li
        float ig, fg, gg, og, c_new, h_new;
        lstm_cell<true>(pre, c_state, ig, fg, gg, og, c_new, h_new);
        const bool active = t < mylen;
        if (active) { c_state = c_new; h_state = h_new; }
        const float hn = row_shl_f<1>(h_state);          // the odd unit of the pair
        if (li < 4) {
            if ((li & 1) == 0) {
                // publish h_t: one {epoch, pair} granule (a row that does not take part publishes its frozen / zero state)
                const unsigned long long gran = ((unsigned long long)(unsigned)(s + 1) << 32) | pack_op16x2(h_state, hn);
                unsigned long long* dst = gbase + (size_t)(s & 1) * 16 * GRAN_F + gran_index(kg, u);
                __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (active) {
                const size_t row = (size_t)t * B + b0 + kg;
                p.y[row * p.ldy + dir * BH + u] = h_state;
                float* gp = p.gates[dir] + row * 4 * BH + u;
                gp[0] = ig; gp[BH] = fg; gp[2 * BH] = gg; gp[3 * BH] = og;
                p.cell[dir][row * BH + u] = c_state;
            }
        }
        gxv = gxn;
    }
    if (dead && lane == 0) atomicExch(p.status, 1);
}

__global__ __launch_bounds__(256) void bilstm_persist_bwd_k(BiBwdP p) {
    extern __shared__ float occupancy_pad[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int xcd, q;
    if (!join(p.census, p.status, tid, xcd, q)) return;
    if (q >= 8) return;                                  // 4 workgroups per (XCD, direction)
    const int dir = q >> 2, m = q & 3;
    const int B = p.B, T = p.T, b0 = xcd * 4;
    const int jt = m * 4 + wave;                         // this wave's column tile: units jt*16 .. jt*16 + 15
    const int u = jt * 16 + li;
    int len[4], tg = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        len[r] = (b0 + r < B) ? min(p.lens[b0 + r], T) : 0;
        tg = len[r] > tg ? len[r] : tg;
    }
    bf16x8 w[32];
    {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.wTfrag[dir]);
#pragma unroll
        for (int c = 0; c < 32; ++c) w[c] = wf[((size_t)jt * 32 + c) * 64 + lane];
    }
    unsigned long long* gbase = p.gran + ((size_t)xcd * 2 + dir) * GRAN_B;          // + parity * 16 * GRAN_B
    __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) rs[par] = __builtin_amdgcn_make_buffer_rsrc(gbase + (size_t)par * 16 * GRAN_B, 0, GRAN_B * 8, 0x00020000);
    const int voff = lane * 16;
    const long t_start = wall_clock64();
    // As in the forward kernel, one shuffle per row hands row r of the MFMA result (register r of lanes 0..15, lane <-> unit) to
    // the lane group kg = r: every lane finishes ONE (row, unit) element -- six row loads, one cell backward, four granule and
    // four dgx stores per lane and step instead of four times as many.
    const int mylen = kg == 0 ? len[0] : kg == 1 ? len[1] : kg == 2 ? len[2] : len[3];
    const float* gp = p.gates[dir];
    const float* cp = p.cell[dir];
    struct Rows { float g[4]; float dy; float cprev; };
    auto load_rows = [&](int t, Rows& v) {               // saved gates i f g o, dy, and the cell of the recurrence's previous step
        const int tp = dir ? t + 1 : t - 1;              // the recurrence's previous step in time
        const bool on = t >= 0 && t < mylen;
        const size_t row = (size_t)t * B + b0 + kg;
#pragma unroll
        for (int g = 0; g < 4; ++g) v.g[g] = on ? gp[row * 4 * BH + g * BH + u] : 0.f;
        v.dy = on ? p.dy[row * p.ldy + dir * BH + u] : 0.f;
        // (also for a row that only joins at the walk's NEXT step: this value becomes its c_t there)
        v.cprev = (tp >= 0 && tp < mylen) ? cp[((size_t)tp * B + b0 + kg) * BH + u] : 0.f;
    };
    // backward walks the recurrence the other way round: forward direction t = tg-1 .. 0, reverse direction t = 0 .. tg-1
    Rows cur, nxt;
    const int t0 = dir ? 0 : tg - 1;
    load_rows(t0, cur);
    float c_t = (t0 >= 0 && t0 < mylen) ? cp[((size_t)t0 * B + b0 + kg) * BH + u] : 0.f, dc_carry = 0.f;
    bool dead = false;
    for (int s = 0; s < tg; ++s) {
        const int t = dir ? s : tg - 1 - s;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            // the group's whole dgates vector (16 KB of tagged granules) in one sweep: 16 x 16-byte loads in flight
            u32x4 ld[8][2];
            if (!poll<8>(ld, rs[(s - 1) & 1], voff, (unsigned)s, t_start, p.timeout_ticks, p.status)) { dead = true; break; }
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const u32x4 pay = (u32x4){ld[c >> 2][0][0], ld[c >> 2][0][2], ld[c >> 2][1][0], ld[c >> 2][1][2]};
                acc = mfma16(__builtin_bit_cast(bf16x8, member(pay, c & 3)), w[c], acc);
            }
        }
        if (s + 1 < tg) load_rows(dir ? t + 1 : t - 1, nxt);
        const float x1 = __shfl(acc[1], li, 64), x2 = __shfl(acc[2], li, 64), x3 = __shfl(acc[3], li, 64);
        const float dh_rec = kg == 0 ? acc[0] : kg == 1 ? x1 : kg == 2 ? x2 : x3;                // (row kg, unit u)
        float da[4], carry;
        lstm_cell_bwd<true>(dh_rec + cur.dy, dc_carry, cur.g[0], cur.g[1], cur.g[2], cur.g[3], c_t, cur.cprev, da, carry);
        const bool active = t < mylen;
        if (active) {
            dc_carry = carry;
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) da[g] = 0.f;
        }
        // publish dgates_s: k = gate * H + unit, one granule per unit pair (even lanes)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float nb = row_shl_f<1>(da[g]);
            if ((li & 1) == 0) {
                const unsigned long long gran = ((unsigned long long)(unsigned)(s + 1) << 32) | pack_op16x2(da[g], nb);
                unsigned long long* dst = gbase + (size_t)(s & 1) * 16 * GRAN_B + gran_index(kg, g * BH + u);
                __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (active) {
            float* dg = p.dgx[dir] + ((size_t)t * B + b0 + kg) * 4 * BH + u;
            dg[0] = da[0]; dg[BH] = da[1]; dg[2 * BH] = da[2]; dg[3 * BH] = da[3];
        }
        c_t = cur.cprev;                                 // the cell of the step the walk visits next
        cur = nxt;
    }
    if (dead && lane == 0) atomicExch(p.status, 1);
}

}  // namespace

static inline size_t al256b(size_t v) { return (v + 255) & ~size_t(255); }
constexpr size_t BI_WFRAG = (size_t)4 * BH * BH * 2;               // one direction's fragment image
constexpr size_t BI_GRAN = (size_t)2 * 16 * GRAN_B * 8;            // the larger (backward) hand-off region
constexpr size_t BI_LDS = 96 * 1024;                               // one workgroup per CU: the census expects 32 per XCD

#if FT_OPFMT == 0
extern "C" int ft_lstm_persist_supported(int B, int H);
extern "C" int ft_bilstm_persist_supported(int B, int H) {
    if (H != BH || B < 1 || B > 32) return 0;
    return ft_lstm_persist_supported(B, 1024);                     // the same 256-CU device test
}
extern "C" size_t ft_bilstm_persist_workspace_bytes(int B, int H) {
    (void)B; (void)H;
    return 2 * al256b(BI_WFRAG) + al256b(BI_GRAN) + 256;
}
#else
extern "C" int ft_bilstm_persist_supported(int B, int H);
#endif

extern "C" int FT_OPNAME(ft_bilstm_persist_fwd)(const float* gx_f, const float* gx_r, const float* w_hh_f, const float* w_hh_r,
                                             const int32_t* lens, float* y, int64_t ldy, float* gates_f, float* gates_r,
                                             float* cell_f, float* cell_r, void* work, int32_t* status, int T, int B, int H,
                                             void* stream) {
    FT_CHECK_ARG(gx_f && gx_r && w_hh_f && w_hh_r && lens && y && gates_f && gates_r && cell_f && cell_r && work && status);
    FT_CHECK_ARG(T >= 0 && ldy >= 2 * (int64_t)H && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    if (!ft_bilstm_persist_supported(B, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_bilstm_persist_fwd: needs H == 256, B <= 32 and a 256-CU device (H=%d B=%d)", H, B);
    if (T == 0) return FT_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(work);
    unsigned short* wf[2] = {reinterpret_cast<unsigned short*>(base), reinterpret_cast<unsigned short*>(base + al256b(BI_WFRAG))};
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(base + 2 * al256b(BI_WFRAG));
    unsigned* census = reinterpret_cast<unsigned*>(base + 2 * al256b(BI_WFRAG) + al256b(BI_GRAN));
    FT_CHECK_HIP(hipMemset2DAsync(y, (size_t)ldy * 4, 0, (size_t)2 * H * 4, (size_t)T * B, st));   // padded frames: zeros
    // tags 0: no epoch matches (epochs start at 1); preset, with the census counters, by the first fragment kernel (WfragAux)
    static_assert(((size_t)2 * 16 * GRAN_F * 8) % 16 == 0, "granule buffer in 16-byte pieces");
    const WfragAux aux{reinterpret_cast<uint4*>(gran), (unsigned long)((size_t)2 * 16 * GRAN_F * 8 / 16), 0u, census};
    hipLaunchKernelGGL(make_wfrag_fwd, dim3(256), dim3(256), 0, st, w_hh_f, wf[0], H, aux);
    hipLaunchKernelGGL(make_wfrag_fwd, dim3(256), dim3(256), 0, st, w_hh_r, wf[1], H, WfragAux{});
    BiFwdP p{{gx_f, gx_r}, lens, y, (long)ldy, {gates_f, gates_r}, {cell_f, cell_r}, {wf[0], wf[1]}, gran, status, census, T, B,
             100000000L / 2};
    FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bilstm_persist_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BI_LDS));
    hipLaunchKernelGGL(bilstm_persist_fwd_k, dim3(NWG), dim3(256), BI_LDS, st, p);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_bilstm_persist_bwd)(const float* dy, int64_t ldy, const float* w_hh_f, const float* w_hh_r, const int32_t* lens,
                                             const float* gates_f, const float* gates_r, const float* cell_f, const float* cell_r,
                                             float* dgx_f, float* dgx_r, void* work, int32_t* status, int T, int B, int H,
                                             void* stream) {
    FT_CHECK_ARG(dy && w_hh_f && w_hh_r && lens && gates_f && gates_r && cell_f && cell_r && dgx_f && dgx_r && work && status);
    FT_CHECK_ARG(T >= 0 && ldy >= 2 * (int64_t)H && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    if (!ft_bilstm_persist_supported(B, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_bilstm_persist_bwd: needs H == 256, B <= 32 and a 256-CU device (H=%d B=%d)", H, B);
    if (T == 0) return FT_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(work);
    unsigned short* wf[2] = {reinterpret_cast<unsigned short*>(base), reinterpret_cast<unsigned short*>(base + al256b(BI_WFRAG))};
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(base + 2 * al256b(BI_WFRAG));
    unsigned* census = reinterpret_cast<unsigned*>(base + 2 * al256b(BI_WFRAG) + al256b(BI_GRAN));
    FT_CHECK_HIP(hipMemsetAsync(dgx_f, 0, (size_t)T * B * 4 * H * 4, st));
    FT_CHECK_HIP(hipMemsetAsync(dgx_r, 0, (size_t)T * B * 4 * H * 4, st));
    static_assert(BI_GRAN % 16 == 0, "granule buffer in 16-byte pieces");
    const WfragAux aux{reinterpret_cast<uint4*>(gran), (unsigned long)(BI_GRAN / 16), 0u, census};
    hipLaunchKernelGGL(make_wfrag_bwd, dim3(256), dim3(256), 0, st, w_hh_f, wf[0], H, aux);
    hipLaunchKernelGGL(make_wfrag_bwd, dim3(256), dim3(256), 0, st, w_hh_r, wf[1], H, WfragAux{});
    BiBwdP p{dy, (long)ldy, lens, {gates_f, gates_r}, {cell_f, cell_r}, {dgx_f, dgx_r}, {wf[0], wf[1]}, gran, status, census, T, B,
             100000000L / 2};
    FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bilstm_persist_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BI_LDS));
    hipLaunchKernelGGL(bilstm_persist_bwd_k, dim3(NWG), dim3(256), BI_LDS, st, p);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
