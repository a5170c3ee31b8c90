// Persistent LSTM recurrences, round 6: ROWS per XCD group as a template parameter, time WINDOWS with carried state, two ROLES per launch.
//
// lstm_persist.hip (rounds 2-5) fixes the geometry at 8 XCD groups x 4 batch rows: an MFMA tile has 16 rows, a group uses 4, and a
// launch holds ONE recurrence over the whole sequence.  The three recurrences of a flow are a chain of 3 x T dependent steps at
// ~1.8 us each, at 5 % of either hardware roof.  This file keeps the step of those kernels -- W_hh as MFMA fragments
// in the accumulation registers of one wave per SIMD, group == XCD (run-time XCC census), hand-offs that never leave the XCD's L2:
// bare operand pairs behind a sentinel (forward), fp32 partials tagged in the mantissa LSB (reduce-scatter backward) -- and makes
// three things parameters:
//   R       batch rows per group, 4 | 8 | 16 (template): B = 64 / 128 in ONE launch per sequence instead of 2 / 4 sliced ones, and
//           B = 32 on FOUR of the eight XCDs;
//   window  a launch covers the time steps [t0, t1) of its sequence; the recurrent state enters and leaves through small fp32
//           buffers (h, c forward; dgates, dc backward), so a sequence can be walked in chunks;
//   roles   a launch carries up to two independent recurrences ("roles"): XCDs 0-3 run role 0, XCDs 4-7 role 1, each on R = 8 rows
//           per XCD for a batch of 32.  With windows this is what lets the two decoder layers of a flow run CONCURRENTLY, layer 1
//           one chunk behind layer 0 (backward: layer 0 one chunk behind layer 1), with layer 1's input projection -- a chip-filling
//           GEMM per chunk -- between the launches (ops.DecoderPairFn): the dependency chain of the pair shrinks from 2 T steps
//           to (1 + 1/n) T steps of the 8-row kernel.
// Launch context (`ctx`, ft_lstm_roles_ctx_bytes): hand-off buffers and census counters live in two SETS per kernel kind; launch i of
// a kind uses set i & 1 and presets set (i + 1) & 1 for its successor while it runs (nobody reads that set during launch i), so a
// launch needs no memset / preset dispatch in front of it.  The caller passes the launch counter of the kind (`phase`).
// Arithmetic: the forward kernel sums exactly like lstm_persist_fwd_k (k-chunks w, w + 4, .. per wave, wave partials 0 + 1 + 2 + 3,
// then gx): bit-identical to it and to the launch-per-step kernel for every R; the backward kernel at R = 4 sums like
// lstm_persist_bwd_rs_k (bit-identical to it), at R = 8 / 16 eight producers per wave in registers, then the four waves.
#include "lstm_persist_common.h"

namespace {

constexpr int CPG = 32, UPC = 32, TPC = 8;            // group == XCD: 32 CUs x 32 hidden units, 8 gate tiles (4 units x 4 gates) per CU
constexpr int RMAX = 16;
constexpr int SBMAX = 32;
constexpr int RRING = 8, RDIST = 6;                   // backward ring (see lstm_persist.hip)

// ---- launch context layout (bytes): [256: census, 2 kinds x 2 sets x 8 counters] [fwd set 0][fwd set 1][bwd set 0][bwd set 1]
constexpr int NBUF = 4;                               // rotating hand-off buffers of the forward kernel (see its publish)
constexpr size_t FWD_SET_DW = (size_t)8 * NBUF * (NCHUNK * 32 * RMAX / 2);       // dwords: 8 groups x 4 buffers x 32 KB (R = 16)
constexpr size_t BWD_SET_FL = (size_t)2 * 8 * 32 * 32 * 32 * RMAX;               // floats: 2 parities x 8 groups x 2 MB (R = 16)
constexpr size_t CTX_BYTES = 256 + 2 * FWD_SET_DW * 4 + 2 * BWD_SET_FL * 4;

struct FwdRoleP {
    const float* gx; const int* lens; float* y; long ldy; float* gates; float* cell;
    const unsigned short* wfrag;         // make_wfrag_fwd_ug image
    float* st_h; float* st_c;            // [B][H]: read when t0 > 0, written at the end (either may be null together)
    int B, LB, t0, t1;
};                                       // (gx: fp32 rows, or 16-bit rows of the build's operand format in the GX16 kernels)
struct FwdLaunchP {
    FwdRoleP role[2];
    int nroles;
    unsigned* hand;                      // ctx + 256
    unsigned* census;                    // ctx (kind 0: counters [set][8])
    int phase, reset_rows;
    int* status; long timeout_ticks; long* prof;
};

template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {                                  // a pointer the caller knows to be wave-uniform, in SGPRs
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void dma16(const float* src, unsigned lds_addr) {       // 1 KiB piece: 16 bytes per lane, LDS = M0 + 16 lane
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr)) : "memory");
}

// preset of the OTHER set of a kind (all-ones: sentinels / tag 1) and of its census counters, spread over the 256 workgroups
__device__ __forceinline__ void preset_other(uint4* other, size_t n16, unsigned* census_other, int wg, int tid) {
    const uint4 v = {SENT, SENT, SENT, SENT};
    for (size_t i = (size_t)wg * 256 + tid; i < n16; i += (size_t)NCU * 256) other[i] = v;
    if (wg == 0 && tid < 8) census_other[tid] = 0u;
}

// GX16: the input-projection rows gx arrive as 16-BIT values of the operand format (written by the projection GEMM's epilogue,
// FT_GEMM_C16) instead of fp32: half the bytes the GEMM writes and the bursts read, twice the steps per burst in the same LDS; the
// value is widened exactly and added to the fp32 recurrent sums like the fp32 row was (round 6; VERDICT r5 #1c)
template <int R, bool PROF, bool GX16>
__global__ __launch_bounds__(256, 1) void lstm_roles_fwd_k(FwdLaunchP P) {
    // Epilogue layout.  R = 4 (the round-5 arrangement): waves 0-1 own one element each and run the cell update, waves 2-3 store the saved
    // tensors of the previous step out of LDS.  R >= 8 (SPLIT): ALL FOUR waves own R / 8 elements each (rows er0 + 8 e) and store their
    // own saved tensors straight from registers behind their publish -- two elements per thread on half the waves put 0.57 us of cell
    // update on the step's critical path where one takes 0.35, and left the other two waves with 12 stores in front of their next poll.
    constexpr bool SPLIT = R >= 8;
    constexpr int ERW = SPLIT ? 8 : 4;             // rows one pass of the epilogue threads covers
    constexpr int EPT = R / ERW;                   // (row, unit) elements per epilogue thread: rows er0 + ERW e
    constexpr int ETH = SPLIT ? 256 : 128;         // epilogue threads
    constexpr int NE = R * UPC;                    // elements per CU
    constexpr int CPL = 16 / R, NLG = 8 / CPL;     // k-chunks per 16-byte load, loads per wave and step
    constexpr int SBF = R == 16 ? 4 : SBMAX * 4 / R;  // steps per fp32 gx burst (64 KB of LDS; 32 KB at R = 16, whose reduce buffers take 80)
    constexpr int SB = GX16 ? 2 * SBF : SBF;          // 16-bit rows: twice the steps in the same bytes
    constexpr int DWG = NCHUNK * 32 * R / 2;       // dwords per hand-off buffer and group
    // LDS: 4-wave reduce (double buffered) | SB steps of gx rows [s][gate][e] | 2 steps of outputs [parity][y,i,f,g,o,c][e]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*red)[4][TPC][R][20] = reinterpret_cast<float (*)[4][TPC][R][20]>(smem);
    float* gxs = smem + 2 * 4 * TPC * R * 20;
    float* outs = gxs + SBF * 4 * NE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int grp, q;
    if (!join_group_local<CPG>(P.census + (P.phase & 1) * 8, P.status, tid, grp, q)) return;
    preset_other(reinterpret_cast<uint4*>(P.hand + (size_t)((P.phase + 1) & 1) * FWD_SET_DW), (size_t)8 * NBUF * (NCHUNK * 32 / 2) * P.reset_rows / 4,
                 P.census + ((P.phase + 1) & 1) * 8, grp * CPG + q, tid);
    const int gpr = 8 / P.nroles;                                // XCD groups per role
    const FwdRoleP p = grp < gpr ? P.role[0] : P.role[1];
    const int B = p.B, LB = p.LB, t0 = p.t0, t1 = p.t1;
    const int b0 = (grp < gpr ? grp : grp - gpr) * R;

    // ---- resident weights: tile j, k-chunk (wave + 4 i), parked in ACCUMULATION registers for the whole launch
    bf16x8 w[TPC][8];
    {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.wfrag);
#pragma unroll
        for (int j = 0; j < TPC; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) w[j][i] = wf[((size_t)(q * TPC + j) * NCHUNK + (wave + 4 * i)) * 64 + lane];
#pragma unroll
        for (int j = 0; j < TPC; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(w[j][i]));
    }

    // ---- epilogue role: thread owns unit el of this CU for the rows er0 + ERW e
    const bool erole = tid < ETH;
    const int el = tid & 31, er0 = (tid >> 5) & (ERW - 1);
    const int eu = q * UPC + el;
    int len[EPT];
    bool ev[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int eb = b0 + er0 + ERW * e;
        ev[e] = erole && eb < B;
        len[e] = ev[e] ? p.lens[eb] : 0;
    }
    int tg = 0;                                                   // steps this group runs: its longest sequence, clipped to the window
    for (int r = 0; r < R; ++r) {
        const int bb = b0 + r;
        if (bb < B) { const int l = p.lens[bb]; tg = l > tg ? l : tg; }
    }
    tg = tg < t1 ? tg : t1;
    tg = __builtin_amdgcn_readfirstlane(tg);

    // ---- gx rows arrive in synchronous bursts of SB steps by 1 KiB LDS-DMA pieces: piece pi of a step = floats [256 pi, 256 pi + 256) of
    // its [gate][row][unit] block (R / 2 pieces per step)
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    auto burst = [&](int t) {
        __syncthreads();
        const int nst = (tg - t) < SB ? (tg - t) : SB;
        const unsigned dst0 = (unsigned)(size_t)(lds_void*)gxs;
        if constexpr (GX16) {
            // 16-bit rows: a step's [gate][row][unit] block is R / 4 pieces of 1 KiB (512 values, eight per lane)
            const unsigned short* gx16 = reinterpret_cast<const unsigned short*>(p.gx);
#pragma unroll
            for (int kk = 0; kk < SB * R / 16; ++kk) {
                const int c = 4 * kk + wu, st = c / (R / 4), pi = c % (R / 4);
                const int f = pi * 512 + lane * 8, gate = f / NE, row = (f % NE) >> 5, unit = f & 31;
                if (st < nst && b0 + row < B)
                    dma16(reinterpret_cast<const float*>(gx16 + (((size_t)(t + st) * LB + b0 + row) * 4 + gate) * PH + q * UPC + unit),
                          dst0 + (unsigned)(st * 4 * NE * 2 + pi * 1024));
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < SB * R / 8; ++kk) {
                const int c = 4 * kk + wu, st = c / (R / 2), pi = c % (R / 2);
                const int f = pi * 256 + lane * 4, gate = f / NE, row = (f % NE) >> 5, unit = f & 31;
                if (st < nst && b0 + row < B)
                    dma16(p.gx + (((size_t)(t + st) * LB + b0 + row) * 4 + gate) * PH + q * UPC + unit, dst0 + (unsigned)((st * 4 * NE + pi * 256) * 4));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // ---- output role (waves 2-3): thread stores the saved tensors of the elements (orow0 + 4 e, ounit) of the previous step
    const int oe0 = tid - 128;
    const int ounit = oe0 & 31, orow0 = (oe0 >> 5) & 3, ou = q * UPC + ounit;
    bool ovalid[EPT];
    int olen[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int ob = b0 + orow0 + 4 * e;
        ovalid[e] = !SPLIT && tid >= 128 && ob < B;
        olen[e] = ovalid[e] ? p.lens[ob] : 0;
    }
    auto store_outputs = [&](int t) {
        if constexpr (SPLIT) return;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            if (!ovalid[e]) continue;
            const float* o = outs + (t & 1) * 6 * NE + oe0 + 128 * e;
            const size_t row = (size_t)t * LB + b0 + orow0 + 4 * e;
            p.y[row * p.ldy + ou] = o[0];
            if (p.gates && t < olen[e]) {
                float* gp = p.gates + row * 4 * PH + ou;
                gp[0] = o[NE]; gp[(size_t)PH] = o[2 * NE]; gp[(size_t)2 * PH] = o[3 * NE]; gp[(size_t)3 * PH] = o[4 * NE];
                p.cell[row * PH + ou] = o[5 * NE];
            }
        }
    };

    float c_state[EPT], h_state[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) { c_state[e] = 0.f; h_state[e] = 0.f; }
    // SPLIT: the saved tensors of a step leave from registers one step later, right BEHIND the next step's poll loads -- vmcnt retires
    // in order and counts stores, so six stores in front of the poll put their issue time and their acknowledgements (~0.3 us) on
    // the step's critical path; behind it they ride under the hop the poll waits for
    float po[EPT][6];
#pragma unroll
    for (int e = 0; e < EPT; ++e)
#pragma unroll
        for (int k = 0; k < 6; ++k) po[e][k] = 0.f;
    // (no branch around the stores: with a path that skips them the compiler's wait for the poll loads in front of them has to
    //  assume the shorter queue, i.e. it waits for the stores as well -- lanes without a row, a window's first step and a launch
    //  without saved tensors aim at a dummy line of the launch context instead.  Rows of finished sequences get their frozen state.)
    float* const dummy = reinterpret_cast<float*>(P.census) + 32 + (lane & 31);
    auto flush_prev = [&](int tp, bool real) {                   // step tp's y / gates / cell of this thread's elements
        if constexpr (!SPLIT) return;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const bool ok = real && ev[e];
            const size_t row = (size_t)tp * LB + b0 + er0 + ERW * e;
            float* yp = ok ? p.y + row * p.ldy + eu : dummy;
            float* gp = ok && p.gates ? p.gates + row * 4 * PH + eu : dummy;
            float* cp = ok && p.gates ? p.cell + row * PH + eu : dummy;
            const size_t gs = ok && p.gates ? (size_t)PH : 0;
            *yp = po[e][0];
            gp[0] = po[e][1]; gp[gs] = po[e][2]; gp[2 * gs] = po[e][3]; gp[3 * gs] = po[e][4];
            *cp = po[e][5];
        }
    };
    unsigned* const bare0 = uniform_ptr(P.hand + (size_t)(P.phase & 1) * FWD_SET_DW + (size_t)grp * NBUF * DWG);
    const __amdgpu_buffer_rsrc_t rbare = __builtin_amdgcn_make_buffer_rsrc(bare0, 0, NBUF * DWG * 4, 0x00020000);
    auto publish = [&](int buf) {                                // h of this thread's elements as operand pairs (even units store)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float h_nb = __uint_as_float(row_shl<1>(__float_as_uint(h_state[e])));
            if ((el & 1) == 0)
                __hip_atomic_store((gu32*)(bare0 + buf * DWG + bare_index<R, 8>(er0 + ERW * e, eu)), pack_op16x2(h_state[e], h_nb),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    int m3 = 0;                                                  // buffer the current step publishes into
    const bool carry = t0 > 0 && tg > t0;
    if (carry) {
        // state of step t0 - 1 from the previous window: published as "step -1" into buffer 0, the window's steps then use 1, 2, 3, 0, ..
        if (erole) {
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (ev[e]) {
                    const size_t o = (size_t)(b0 + er0 + ERW * e) * PH + eu;
                    h_state[e] = p.st_h[o]; c_state[e] = p.st_c[o];
                }
            publish(0);
        }
        m3 = 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const int voff = lane * 16;
    const int soff_w = wu * (NLG * 1024);
    const long t_start = wall_clock64();
    bool dead = false;
    const bool prof = PROF && P.prof != nullptr && grp == 0 && q == 0 && lane == 0;

    for (int t = t0; t < tg; ++t) {
        const int n = t - t0;
        if ((n % SB) == 0) burst(t);
        long st0 = 0, st1 = 0, st2 = 0, st3 = 0, npass = 0;
        if (prof) st0 = wall_clock64();
        f32x4 acc[TPC];
#pragma unroll
        for (int j = 0; j < TPC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // zeroed HERE, in front of the gather: otherwise the compiler sinks each zeroing move to right in front of the inline-asm MFMA that
        // first reads the register -- a VALU write -> MFMA SrcC hazard it does not pad for an asm statement
#pragma unroll
        for (int j = 0; j < TPC; ++j) asm volatile("" : "+v"(acc[j]));
        if (n > 0 || carry) {
            const int boff = __builtin_amdgcn_readfirstlane(((m3 + NBUF - 1) % NBUF) * (DWG * 4));  // the previous step's buffer (scalar offset)
            u32x4 ld[NLG];
#pragma unroll
            for (int g = 0; g < NLG; ++g) ld[g] = __builtin_amdgcn_raw_buffer_load_b128(rbare, voff, soff_w + g * 1024 + boff, 2);
            __builtin_amdgcn_sched_barrier(0);                     // (the scheduler otherwise hoists the first poll compare above the stores)
            flush_prev(t - 1, n > 0);
            __builtin_amdgcn_sched_barrier(0);
            unsigned ready = 0;
            auto check = [&]() {
#pragma unroll
                for (int g = 0; g < NLG; ++g) {
                    if (!((ready >> g) & 1u)) {
                        if (__all((ld[g][0] != SENT) & (ld[g][1] != SENT) & (ld[g][2] != SENT) & (ld[g][3] != SENT))) ready |= 1u << g;
                    }
                }
            };
            // the FIRST pass in straight-line code: the compiler then waits for exactly the poll loads (vmcnt = the stores issued behind
            // them); at the head of the spin loop it can only wait for everything in flight
            check();
            if (PROF) npass = 1;
            if (ready != (1u << NLG) - 1u) {
                for (unsigned spins = 0;; ++spins) {
                    if ((spins & 15) == 15) {
                        const int st_now = __builtin_amdgcn_readfirstlane(__hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        if (wall_clock64() - t_start > P.timeout_ticks || st_now != 0) { dead = true; break; }
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int g = 0; g < NLG; ++g) {
                        if (!((ready >> g) & 1u)) ld[g] = __builtin_amdgcn_raw_buffer_load_b128(rbare, voff, soff_w + g * 1024 + boff, 2);
                    }
                    check();
                    if (PROF) npass = spins + 2;
                    if (ready == (1u << NLG) - 1u) break;
                }
            }
            if (dead) break;
            // chunk i of this wave = load i / CPL, member i % CPL: the four moves that put a chunk's rows into the MFMA row positions (DPP row
            // shift by (i % CPL) R lanes) are issued one per MFMA gap of the PREVIOUS chunk, three operand buffers in rotation
            auto operand_word = [&](int i, int c) -> unsigned {
                const unsigned v = ld[i / CPL][c];
                switch (i % CPL) {
                    case 0: {       // (an explicit move IN this slot: a plain copy is materialised by the compiler right in front of the
                        unsigned r; //  consuming asm MFMA -- VALU write -> MFMA source read without the two wait states)
                        asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
                        return r;
                    }
                    case 1: return row_shl<R & 15>(v);
                    case 2: return row_shl<(2 * R) & 15>(v);
                    default: return row_shl<(3 * R) & 15>(v);
                }
            };
            u32x4 au[3];
#pragma unroll
            for (int c = 0; c < 4; ++c) au[0][c] = operand_word(0, c);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < TPC; ++j) {
                    if (i == 0 && j == 0) mfma16_bagpr_nop(acc[0], au[0], __builtin_bit_cast(u32x4, w[0][0]));
                    else mfma16_bagpr(acc[j], au[i % 3], __builtin_bit_cast(u32x4, w[j][i]));
                    if (i + 1 < 8 && j < 4) {
                        __builtin_amdgcn_sched_barrier(0);
                        au[(i + 1) % 3][j] = operand_word(i + 1, j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");     // asm MFMAs: 12 wait states before anything reads the last one's result
        }
        // D[m = batch row kg * 4 + r][n = li]: rows >= R are padding
        const int rb = n & 1;
        if (prof) st1 = wall_clock64();
        if (kg * 4 < R) {
#pragma unroll
            for (int j = 0; j < TPC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[rb][wave][j][kg * 4 + r][li] = acc[j][r];
        }
        __syncthreads();
        if (prof) st2 = wall_clock64();
        if (n > 0) store_outputs(t - 1);
        if (erole) {
            const int j = el >> 2, ul = el & 3;
            float ig[EPT], fg[EPT], gg[EPT], og[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int row = er0 + ERW * e;
                const float* gxr = gxs + (n % SB) * 4 * NE + row * UPC + el;
                const unsigned short* gxr16 = reinterpret_cast<const unsigned short*>(gxs) + (n % SB) * 4 * NE + row * UPC + el;
                // the four gates of (unit, row) are adjacent (make_wfrag_fwd_ug): one 16-byte read per wave partial; the sum keeps its
                // order (wave 0 + 1 + 2 + 3, then gx)
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(&red[rb][0][j][row][ul * 4]), p1 = *reinterpret_cast<const f32x4*>(&red[rb][1][j][row][ul * 4]);
                const f32x4 p2 = *reinterpret_cast<const f32x4*>(&red[rb][2][j][row][ul * 4]), p3 = *reinterpret_cast<const f32x4*>(&red[rb][3][j][row][ul * 4]);
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = p0[g] + p1[g] + p2[g] + p3[g] + (GX16 ? op16_to_f(gxr16[g * NE]) : gxr[g * NE]);
                ig[e] = fg[e] = gg[e] = og[e] = 0.f;
                if (t < len[e]) {
                    float c_new, h_new;
                    lstm_cell<true>(pre, c_state[e], ig[e], fg[e], gg[e], og[e], c_new, h_new);
                    c_state[e] = c_new; h_state[e] = h_new;
                }
            }
            // Sentinel protocol over FOUR rotating buffers (round 6; three through round 5).  Step t publishes into buffer m3 and then
            // resets this thread's slots of buffer m3 + 2 -- it holds step t - 2, which every consumer has finished reading: they all
            // published step t - 1, and all four waves of this CU had gathered all of it when they passed the barrier above.  A
            // consumer polls that buffer for step t + 2, i.e. after it has seen this CU's publish of step t + 1 -- and THAT publish
            // waits (vmcnt(0) below, one step from now) for this reset's acknowledgement, which by then is a whole step old.  With three
            // buffers the reset had to be acknowledged before the publish of the SAME step: a store round trip (~0.1-0.2 us) on the
            // critical path of every step.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the previous step's reset is in the L2 before this publish leaves
            publish(m3);
            if ((el & 1) == 0) {
                const int rst = (m3 + 2) % NBUF;
#pragma unroll
                for (int e = 0; e < EPT; ++e)
                    __hip_atomic_store((gu32*)(bare0 + rst * DWG + bare_index<R, 8>(er0 + ERW * e, eu)), SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (prof) st3 = wall_clock64();
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                if constexpr (SPLIT) {
                    // saved tensors: kept in registers and stored at the top of the NEXT step, behind its poll loads (flush_prev)
                    po[e][0] = t < len[e] ? h_state[e] : 0.f;
                    po[e][1] = ig[e]; po[e][2] = fg[e]; po[e][3] = gg[e]; po[e][4] = og[e]; po[e][5] = c_state[e];
                } else {
                    float* o = outs + (t & 1) * 6 * NE + tid + 128 * e;   // waves 2-3 store it during the next step
                    o[0] = t < len[e] ? h_state[e] : 0.f;
                    o[NE] = ig[e]; o[2 * NE] = fg[e]; o[3 * NE] = gg[e]; o[4 * NE] = og[e]; o[5 * NE] = c_state[e];
                }
            }
        }
        if (PROF) {
            if (prof && n < 1024) {
                long* o = P.prof + ((size_t)n * 4 + wave) * 5;
                o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = npass;
            }
        }
        m3 = (m3 + 1) % NBUF;
    }
    if (dead) {
        if (lane == 0) atomicExch(P.status, 1);
        return;
    }
    __syncthreads();
    if (tg > t0) { store_outputs(tg - 1); flush_prev(tg - 1, true); }
    if (erole) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            if (!ev[e]) continue;
            const int eb = b0 + er0 + ERW * e;
            // pad rows beyond the group's longest sequence inside the window: y = 0 (pad_packed_sequence semantics)
            for (int t = tg > t0 ? tg : t0; t < t1; ++t) p.y[((size_t)t * LB + eb) * p.ldy + eu] = 0.f;
            if (p.st_h && tg > t0) { p.st_h[(size_t)eb * PH + eu] = h_state[e]; p.st_c[(size_t)eb * PH + eu] = c_state[e]; }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Backward recurrence in reduce-scatter form (see lstm_persist_bwd_rs_k for the derivation), R rows per group.
//   partial buffers of a group: [2 parity][consumer 32][producer 32][32 units][R rows] fp32; tags = mantissa LSB of the first and
//   last dword of every 16-byte piece (one lane's store = four rows of one unit).
// Step counter m = n + (1 with a carried-in state): buffer m & 1, tag (m >> 1) & 1; with a carry the launch starts with a "step -1"
// (m = 0) that only multiplies the carried dgates and publishes.
struct BwdRoleP {
    const float* dy; long ldy; const int* lens; const float* gates; const float* cell; float* dgx;
    const unsigned short* wTfrag;        // make_wfrag_rs image
    unsigned short* dimg; long dimg_ld; int dimg_rows; float* dbias;        // optional compact 16-bit image of dgates (+ column sums)
    float* st_da; float* st_dc;          // [B][4H], [B][H]: read with carry, written at the end
    int B, LB, t0, t1, carry;
};
struct BwdLaunchP {
    BwdRoleP role[2];
    int nroles;
    float* part;                         // ctx + 256 + 2 FWD sets
    unsigned* census;                    // ctx + 64 (kind 1)
    int phase, reset_rows;
    int* status; long timeout_ticks; long* prof;
};

template <int R, bool PROF>
__global__ __launch_bounds__(256, 1) void lstm_roles_bwd_k(BwdLaunchP P) {
    // (epilogue layout as in the forward kernel: R = 4 the round-5 arrangement, R >= 8 all four waves run the cell backward of R / 8
    //  elements each and store their own dgates rows / image entries straight from registers)
    constexpr bool SPLIT = R >= 8;
    constexpr int ERW = SPLIT ? 8 : 4, EPT = R / ERW, ETH = SPLIT ? 256 : 128, NE = R * UPC;
    constexpr int NT = PH / 16 / 4;                // column tiles per wave: 16
    constexpr int NPART = R == 4 ? 8 : 4;          // partial sums per element in LDS: (wave, lane half) at R = 4, waves otherwise
    constexpr int NLD = R;                         // 1 KiB gather loads per wave and step
    constexpr int NPS = 6 * NE / 256;              // 1 KiB pieces per ring slot: 3 | 6 | 12
    constexpr int NPS2 = NPS * 2 / 3;              // wave 2 moves pieces [0, NPS2): the gate rows; wave 3 the dy and previous-cell rows
    constexpr size_t PBG = (size_t)CPG * CPG * UPC * R;                      // floats per group and parity
    constexpr int CONS = CPG * UPC * R * 4;        // bytes of a consumer block
    auto tagged = [](int r) { return r == 0 || r == 3; };
    // LDS: gather sums [NPART][32 units][R rows] | dgates operands [4 gates][16 A-tile rows][32 units] 16-bit (rows >= R zero for good)
    // | RING steps in [slot][gates x4, dy, previous cell][e] | 2 steps out [parity][4][e]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gsum = smem;
    unsigned* daop = reinterpret_cast<unsigned*>(smem + NPART * UPC * R);
    float* ins = smem + NPART * UPC * R + 1024;
    float* outs = ins + RRING * 6 * NE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int grp, q;
    if (!join_group_local<CPG>(P.census + (P.phase & 1) * 8, P.status, tid, grp, q)) return;
    preset_other(reinterpret_cast<uint4*>(P.part + (size_t)((P.phase + 1) & 1) * BWD_SET_FL), (size_t)2 * 8 * 32 * 32 * 32 * P.reset_rows / 4,
                 P.census + ((P.phase + 1) & 1) * 8, grp * CPG + q, tid);
    const int gpr = 8 / P.nroles;
    const BwdRoleP p = grp < gpr ? P.role[0] : P.role[1];
    const int lgrp = grp < gpr ? grp : grp - gpr;
    const int B = p.B, LB = p.LB, t0 = p.t0;
    const int b0 = lgrp * R;
    for (int i2 = tid; i2 < 1024; i2 += 256) daop[i2] = 0u;

    // ---- resident weights: tile 16 wave + j, chunk (= gate) g
    u32x4 w[NT][4];
    {
        const u32x4* wf = reinterpret_cast<const u32x4*>(p.wTfrag);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) w[j][g] = wf[((size_t)((q * (PH / 16) + wave * NT + j) * 4 + g)) * 64 + lane];
    }

    const bool erole = tid < ETH;
    const int el = tid & 31, er0 = (tid >> 5) & (ERW - 1);
    const int eu = q * UPC + el;
    int len[EPT];
    bool ev[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int eb = b0 + er0 + ERW * e;
        ev[e] = erole && eb < B;
        len[e] = ev[e] ? p.lens[eb] : 0;
    }
    int te = 0;                                                   // the window's steps of this group: s = te - 1 down to t0
    for (int r = 0; r < R; ++r) {
        const int bb = b0 + r;
        if (bb < B) { const int l = p.lens[bb]; te = l > te ? l : te; }
    }
    te = te < p.t1 ? te : p.t1;
    te = __builtin_amdgcn_readfirstlane(te);
    const int NS = te > t0 ? te - t0 : 0;                         // steps
    const bool carry = p.carry != 0 && NS > 0;

    // ---- output role: waves 2-3 (R = 4: the previous step's rows out of LDS) / every thread for its own elements (SPLIT)
    const int oe0 = SPLIT ? tid : tid - 128;
    const int ounit = oe0 & 31, orow0 = (oe0 >> 5) & (ERW - 1), ou = q * UPC + ounit;
    bool ovalid[EPT];
    int olen[EPT], ooff[EPT];
    float bsum[EPT][4];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int ob = b0 + orow0 + ERW * e;
        ovalid[e] = (SPLIT || tid >= 128) && ob < B;
        olen[e] = 0; ooff[e] = 0;
        if (p.dimg && ovalid[e]) {
            olen[e] = p.lens[ob];
            for (int bb = 0; bb < ob; ++bb) ooff[e] += p.lens[bb] + 1;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) bsum[e][g] = 0.f;
    }
    auto emit = [&](int e, int so, float v0, float v1, float v2, float v3) {     // dgates of element e at time so: fp32 rows and / or image
        if (p.dgx) {
            float* dg = p.dgx + ((size_t)so * LB + b0 + orow0 + ERW * e) * 4 * PH + ou;
            dg[0] = v0; dg[(size_t)PH] = v1; dg[(size_t)2 * PH] = v2; dg[(size_t)3 * PH] = v3;
        }
        if (p.dimg && so < olen[e]) {
            unsigned short* ip = p.dimg + (size_t)(ooff[e] + so) * p.dimg_ld + ou;
            ip[0] = (unsigned short)pack_op16x2(v0, 0.f); ip[PH] = (unsigned short)pack_op16x2(v1, 0.f);
            ip[2 * PH] = (unsigned short)pack_op16x2(v2, 0.f); ip[3 * PH] = (unsigned short)pack_op16x2(v3, 0.f);
            bsum[e][0] += v0; bsum[e][1] += v1; bsum[e][2] += v2; bsum[e][3] += v3;
        }
    };
    auto store_outputs = [&](int n) {                            // R = 4: dgates of step counter n (time te - 1 - n) from outs[n & 1]
        if constexpr (SPLIT) return;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            if (!ovalid[e]) continue;
            const float* o = outs + (n & 1) * 4 * NE + oe0 + 128 * e;
            emit(e, te - 1 - n, o[0], o[NE], o[2 * NE], o[3 * NE]);
        }
    };
    // ---- ring fill: slot = 6 rows x NE floats = NPS pieces of 1 KiB; piece k = floats [256 k, 256 k + 256) of the slot: slot row
    // rs = f / NE (gates i f g o, dy, cell of the step before), batch row (f % NE) >> 5, units f & 31 .. + 3
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const unsigned ring0 = (unsigned)(size_t)(lds_void*)ins;
    auto piece_src = [&](int k, int sm, bool& ok) -> const float* {
        const int f = k * 256 + lane * 4, rs = f / NE, row = (f % NE) >> 5, unit = f & 31;
        const size_t r = (size_t)sm * LB + b0 + row;
        ok = b0 + row < B && (rs != 5 || sm > 0);
        if (rs < 4) return p.gates + (r * 4 + rs) * PH + q * UPC + unit;
        if (rs == 4) return p.dy + r * p.ldy + q * UPC + unit;
        return p.cell + (r - LB) * PH + q * UPC + unit;
    };
    auto prefetch = [&](int m) {
        if (wu < 2 || m >= NS) return;
        const int sm = te - 1 - m;
        const unsigned dst = ring0 + (unsigned)((m % RRING) * 6 * NE * 4);
        const int k0 = wu == 2 ? 0 : NPS2, k1 = wu == 2 ? NPS2 : NPS;
#pragma unroll
        for (int k = 0; k < NPS; ++k) {
            if (k < k0 || k >= k1) continue;
            bool ok;
            const float* src = piece_src(k, sm, ok);
            if (ok) dma16(src, dst + k * 1024);
        }
    };
    // the cell of the window's LAST step (c_t of n = 0) goes to row 5 of slot RING - 1
    if (wu == 3 && NS > 0) {
#pragma unroll
        for (int k = NPS2; k < NPS; ++k) {
            const int f = k * 256 + lane * 4, rs = f / NE, row = (f % NE) >> 5, unit = f & 31;
            if (rs == 5 && b0 + row < B)
                dma16(p.cell + ((size_t)(te - 1) * LB + b0 + row) * PH + q * UPC + unit, ring0 + (unsigned)((RRING - 1) * 6 * NE * 4 + k * 1024));
        }
    }
#pragma unroll
    for (int m = 0; m < RDIST; ++m) prefetch(m);

    float dc_carry[EPT], da[EPT][4];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        dc_carry[e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) da[e][g] = 0.f;
    }
    if (carry && erole) {
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (ev[e]) {
                const size_t eb = (size_t)(b0 + er0 + ERW * e);
                dc_carry[e] = p.st_dc[eb * PH + eu];
#pragma unroll
                for (int g = 0; g < 4; ++g) da[e][g] = p.st_da[(eb * 4 + g) * PH + eu];
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and the compiler is TOLD so (see lstm_persist_bwd_rs_k): otherwise the first use of every register loaded above sits inside the
    // loop behind a vmcnt(0) that waits for the ring DMAs and output stores of waves 2-3
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+a"(w[j][g]));
#pragma unroll
    for (int e = 0; e < EPT; ++e) asm volatile("" :: "v"(olen[e]), "v"(ooff[e]), "v"(len[e]), "v"(dc_carry[e]));
    __syncthreads();
    auto write_daop = [&]() {                                   // dgates as MFMA A operands: daop[gate][row][unit pair] (the odd unit from lane + 1)
#pragma unroll
        for (int e = 0; e < EPT; ++e)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float nb = __uint_as_float(row_shl<1>(__float_as_uint(da[e][g])));
                if ((el & 1) == 0) daop[(g * 16 + er0 + ERW * e) * (UPC / 2) + (el >> 1)] = pack_op16x2(da[e][g], nb);
            }
    };
    if (carry) {
        if (erole) write_daop();
        __syncthreads();
    }

    // ONE resource per direction over both parity buffers of the group, the parity selected by a scalar byte offset (and the base made
    // wave-uniform by hand: a descriptor the compiler takes for divergent is applied lane by lane -- a waterfall loop per access)
    float* const gbase = uniform_ptr(P.part + (size_t)(P.phase & 1) * BWD_SET_FL + (size_t)grp * PBG);
    constexpr int PAROFF = (int)(8 * PBG * 4);                         // bytes between the two parity buffers of a group
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gbase + (size_t)q * CPG * UPC * R, 0, PAROFF + CONS, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, PAROFF + (int)(PBG * 4), 0x00020000);
    const int voff = lane * 16;
    const int soff_w = wu * (NLD * 1024);
    const int wvoff = kg * 4 < R ? (q * UPC * R + li * R + kg * 4) * 4 : (int)0x7ffffff0;     // (beyond num_records: dropped)
    const int wsoff_w = wu * 8 * CONS;
    const long t_start = wall_clock64();
    bool dead = false;
    const bool prof = PROF && P.prof != nullptr && grp == 0 && q == 0 && lane == 0;
    const int moff = carry ? 1 : 0;

    for (int n = carry ? -1 : 0; n < NS; ++n) {
        long st0 = 0, st1 = 0, st2 = 0, st3 = 0, npass = 0;
        if (prof) st0 = wall_clock64();
        const int s = te - 1 - n;
        const int m = n + moff;
        if (n >= 0) {
            // everything of the cell backward that does not need dh_rec is evaluated while the gather loads are in flight:
            //   dc = dh fA + dc_carry ; carry' = dc f ; da_i = dc fI ; da_f = dc fF ; da_g = dc fG ; da_o = dh fO
            float fA[EPT], fO[EPT], fI[EPT], fF[EPT], fG[EPT], fgate[EPT], dy_s[EPT];
            bool active[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) { fA[e] = fO[e] = fI[e] = fF[e] = fG[e] = fgate[e] = dy_s[e] = 0.f; active[e] = erole && s < len[e]; }
            auto precompute = [&]() {
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    if (!active[e]) continue;
                    const int ei = tid + ETH * e;
                    const float* in = ins + (n % RRING) * 6 * NE + ei;
                    const float ig = in[0], fg = in[NE], gg = in[2 * NE], og = in[3 * NE];
                    dy_s[e] = in[4 * NE];
                    const float c_t = ins[(((n + RRING - 1) % RRING) * 6 + 5) * NE + ei], c_prev = s > 0 ? in[5 * NE] : 0.f;
                    const float tc = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * 1.4426950408889634f * c_t) + 1.f);
                    fA[e] = og * __fmaf_rn(-tc, tc, 1.f);
                    fO[e] = tc * og * (1.f - og);
                    fI[e] = gg * ig * (1.f - ig);
                    fF[e] = c_prev * fg * (1.f - fg);
                    fG[e] = ig * __fmaf_rn(-gg, gg, 1.f);
                    fgate[e] = fg;
                }
            };
            if (m > 0) {
                // ---- gather: the partials of step m - 1 addressed to this CU
                const unsigned tag = (unsigned)((m - 1) >> 1) & 1u;
                const int roff = soff_w + __builtin_amdgcn_readfirstlane((m - 1) & 1) * PAROFF;
                u32x4 ld[NLD];
#pragma unroll
                for (int g = 0; g < NLD; ++g) ld[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, roff + g * 1024, 2);
                precompute();
                unsigned ready = 0;
                for (unsigned spins = 0;; ++spins) {
#pragma unroll
                    for (int g = 0; g < NLD; ++g) {
                        if (!((ready >> g) & 1u)) {
                            if (__all((((ld[g][0] ^ tag) | (ld[g][3] ^ tag)) & 1u) == 0u)) ready |= 1u << g;
                        }
                    }
                    if (ready == (1u << NLD) - 1u) break;
                    if (PROF) ++npass;
                    if ((spins & 15) == 15) {
                        const int st_now = __builtin_amdgcn_readfirstlane(__hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        if (wall_clock64() - t_start > P.timeout_ticks || st_now != 0) { dead = true; break; }
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int g = 0; g < NLD; ++g) {
                        if (!((ready >> g) & 1u)) ld[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, roff + g * 1024, 2);
                    }
                }
                // every poll load has returned (its data was just compared; the time-out path leaves with some in flight, which costs it
                // nothing to wait for): said HERE, in front of the time-out exit, in a form the compiler's wait-count pass sees -- the exit
                // shares blocks with the carried-in "step -1", which enters the MFMA block without passing the step's front half
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (dead) break;
                auto val = [&](int g, int r) { return __uint_as_float(tagged(r) ? (ld[g][r] & ~1u) : ld[g][r]); };
                if constexpr (R == 4) {
                    // lane (half = lane >> 5, unit = lane & 31) holds rows 0 .. 3 of producers 8 w + 2 g + half
                    f32x4 sacc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[r] = ((val(0, r) + val(1, r)) + val(2, r)) + val(3, r);
                    *reinterpret_cast<f32x4*>(gsum + ((wave * 2 + (lane >> 5)) * UPC + (lane & 31)) * 4) = sacc;
                } else if constexpr (R == 8) {
                    // load g = producer 8 w + g: lane (unit = lane >> 1, rows 4 (lane & 1) ..)
                    f32x4 sacc;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sacc[r] = ((val(0, r) + val(1, r)) + (val(2, r) + val(3, r))) + ((val(4, r) + val(5, r)) + (val(6, r) + val(7, r)));
                    *reinterpret_cast<f32x4*>(gsum + (wave * UPC + (lane >> 1)) * 8 + (lane & 1) * 4) = sacc;
                } else {
                    // loads 2 pp, 2 pp + 1 = producer 8 w + pp, units 0-15 / 16-31: lane (unit = lane >> 2 (+ 16), rows 4 (lane & 3) ..)
                    f32x4 s0, s1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s0[r] = ((val(0, r) + val(2, r)) + (val(4, r) + val(6, r))) + ((val(8, r) + val(10, r)) + (val(12, r) + val(14, r)));
                        s1[r] = ((val(1, r) + val(3, r)) + (val(5, r) + val(7, r))) + ((val(9, r) + val(11, r)) + (val(13, r) + val(15, r)));
                    }
                    *reinterpret_cast<f32x4*>(gsum + (wave * UPC + (lane >> 2)) * 16 + (lane & 3) * 4) = s0;
                    *reinterpret_cast<f32x4*>(gsum + (wave * UPC + 16 + (lane >> 2)) * 16 + (lane & 3) * 4) = s1;
                }
            } else {
                precompute();
            }
            if (prof) st1 = wall_clock64();
            // every poll load has returned; said in a form the compiler's wait-count pass sees on every path into the step body
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();                                         // (1) gather sums visible; dgates operands of the last step consumed
            if (n > 0) store_outputs(n - 1);
            prefetch(n + RDIST);
            if (erole) {
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) da[e][g] = 0.f;
                    if (active[e]) {
                        float dh = dy_s[e];
                        if (m > 0) {
                            const float* gs = gsum + el * R + er0 + ERW * e;
                            constexpr int PS = UPC * R;
                            if constexpr (R == 4) dh += ((gs[0] + gs[PS]) + (gs[2 * PS] + gs[3 * PS])) + ((gs[4 * PS] + gs[5 * PS]) + (gs[6 * PS] + gs[7 * PS]));
                            else dh += (gs[0] + gs[PS]) + (gs[2 * PS] + gs[3 * PS]);
                        }
                        const float dc = __fmaf_rn(dh, fA[e], dc_carry[e]);
                        dc_carry[e] = dc * fgate[e];
                        da[e][0] = dc * fI[e]; da[e][1] = dc * fF[e]; da[e][2] = dc * fG[e]; da[e][3] = dh * fO[e];
                    }
                }
                write_daop();
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    if constexpr (SPLIT) {
                        if (ovalid[e]) emit(e, s, da[e][0], da[e][1], da[e][2], da[e][3]);
                    } else {
                        float* o = outs + (n & 1) * 4 * NE + tid + 128 * e;
#pragma unroll
                        for (int g = 0; g < 4; ++g) o[g * NE] = da[e][g];
                    }
                }
            }
            __syncthreads();                                         // (2) the group's rows of dgates_s are in LDS
            if (prof) st2 = wall_clock64();
        }
        if (n + 1 < NS) {
            // ---- partial dh_rec of the NEXT step: this CU's dgates x its 128 rows of W_hh; rows >= R of the A tile are zero
            u32x4 a[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = *reinterpret_cast<const u32x4*>(daop + (g * 16 + li) * (UPC / 2) + kg * 4);
            // tiles in groups of four; a group's results are tagged and stored while the NEXT group's MFMAs occupy the matrix pipe: ONE
            // filler per MFMA gap.  D rows kg * 4 .. + 3 of column li; tile j of wave w = columns (16 w + j) 16 + li = consumer
            // 8 w + (j >> 1), unit (j & 1) 16 + li: one tagged 16-byte store per tile and lane of the first R / 4 lane rows
            const unsigned tagw = (unsigned)(m >> 1) & 1u;
            const int woff = wsoff_w + __builtin_amdgcn_readfirstlane(m & 1) * PAROFF;
            auto tag_word = [&](f32x4& v, int r) {
                float x = v[r];
                asm volatile("v_and_or_b32 %0, %0, -2, %1" : "+v"(x) : "v"(tagw));
                v[r] = x;
            };
            auto store_tile = [&](int j, const f32x4& acc) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), wrs, wvoff, woff + (j >> 1) * CONS + (j & 1) * (16 * R * 4), 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto filler = [&](int mm, int ptq, f32x4 (&pacc)[4]) {
                const int f = mm >> 2, sl = mm & 3;
                if (sl == 0) tag_word(pacc[f], 0);
                else if (sl == 1) tag_word(pacc[f], 3);
                else if (sl == 3) store_tile(ptq * 4 + f, pacc[f]);
            };
            auto group = [&](int tq, f32x4 (&acc)[4], int ptq, f32x4 (&pacc)[4]) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (g == 0) mfma16_bagpr_first(acc[jj], a[0], w[tq * 4 + jj][0]);
                        else mfma16_bagpr(acc[jj], a[g], w[tq * 4 + jj][g]);
                        if (ptq >= 0) filler(g * 4 + jj, ptq, pacc);
                    }
            };
            f32x4 accA[4], accB[4];
            group(0, accA, -1, accB);
            group(1, accB, 0, accA);
            group(2, accA, 1, accB);
            group(3, accB, 2, accA);
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
            for (int mm = 0; mm < 16; ++mm) filler(mm, 3, accB);
        }
        if (PROF) {
            if (prof) st3 = wall_clock64();
            if (prof && n >= 0 && n < 1024) {
                long* o = P.prof + ((size_t)n * 4 + wave) * 5;
                o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = npass;
            }
        }
    }
    if (dead) {
        if (lane == 0) atomicExch(P.status, 1);
        return;
    }
    __syncthreads();
    if (NS > 0) store_outputs(NS - 1);
    if (erole) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            if (!ev[e]) continue;
            const size_t eb = (size_t)(b0 + er0 + ERW * e);
            if (p.dgx) {                                                 // pad rows beyond the group's longest sequence inside the window
                for (int t = te > t0 ? te : t0; t < p.t1; ++t) {
                    float* dg = p.dgx + ((size_t)t * LB + eb) * 4 * PH + eu;
#pragma unroll
                    for (int g = 0; g < 4; ++g) dg[(size_t)g * PH] = 0.f;
                }
            }
            if (p.st_da && NS > 0) {
                p.st_dc[eb * PH + eu] = dc_carry[e];
#pragma unroll
                for (int g = 0; g < 4; ++g) p.st_da[(eb * 4 + g) * PH + eu] = da[e][g];
            }
        }
    }
    if (p.dimg) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            if (!ovalid[e]) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) atomicAdd(p.dbias + g * PH + ou, bsum[e][g]);
            if (t0 == 0) {                                               // the utterance's zero separator row (once: by the window that holds step 0)
                unsigned short* ip = p.dimg + (size_t)(ooff[e] + olen[e]) * p.dimg_ld + ou;
#pragma unroll
                for (int g = 0; g < 4; ++g) ip[g * PH] = 0;
            }
        }
        if (t0 == 0) {
            // zero rows behind the last utterance up to ceil256(R + 32) (gemm_bf16.hip mapped_rows), spread over the role's workgroups
            int Rr = 0;
            for (int bb = 0; bb < B; ++bb) Rr += p.lens[bb] + 1;
            int Rz = (Rr + 32 + 255) & ~255;
            Rz = Rz < p.dimg_rows ? Rz : p.dimg_rows;
            for (int r = Rr + lgrp * CPG + q; r < Rz; r += gpr * CPG) {
                uint4* row = reinterpret_cast<uint4*>(p.dimg + (size_t)r * p.dimg_ld);
                for (int c = tid; c < 4 * PH / 8; c += 256) row[c] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
}

}  // namespace

#if FT_OPFMT == 0
long* ftint_roles_prof = nullptr;                    // shared with the fp16 build of this file
extern "C" int ft_lstm_roles_debug_prof(void* dev_buf) { ftint_roles_prof = reinterpret_cast<long*>(dev_buf); return FT_OK; }
extern "C" size_t ft_lstm_roles_ctx_bytes(void) { return CTX_BYTES; }
extern "C" size_t ft_lstm_roles_wimg_bytes(int H) { return (size_t)4 * H * H * 2; }
extern "C" int ft_lstm_roles_ctx_init(void* ctx, void* stream) {
    FT_CHECK_ARG(ctx && reinterpret_cast<uintptr_t>(ctx) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    FT_CHECK_HIP(hipMemsetAsync(ctx, 0, 256, st));
    FT_CHECK_HIP(hipMemsetAsync(reinterpret_cast<char*>(ctx) + 256, 0xFF, CTX_BYTES - 256, st));
    return FT_OK;
}
#else
extern long* ftint_roles_prof;
#endif

extern "C" int ft_lstm_persist_supported(int B, int H);

extern "C" int FT_OPNAME(ft_lstm_roles_prepare_fwd)(const float* w_hh, void* wimg, int H, void* stream) {
    FT_CHECK_ARG(w_hh && wimg && H == PH && reinterpret_cast<uintptr_t>(wimg) % 256 == 0);
    hipLaunchKernelGGL(make_wfrag_fwd_ug, dim3(2048), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_hh,
                       reinterpret_cast<unsigned short*>(wimg), H, WfragAux{nullptr, 0ul, 0u, nullptr});
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int FT_OPNAME(ft_lstm_roles_prepare_bwd)(const float* w_hh, void* wimg, int H, void* stream) {
    FT_CHECK_ARG(w_hh && wimg && H == PH && reinterpret_cast<uintptr_t>(wimg) % 256 == 0);
    hipLaunchKernelGGL(make_wfrag_rs, dim3(2048), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_hh,
                       reinterpret_cast<unsigned short*>(wimg), H, WfragAux{nullptr, 0ul, 0u, nullptr});
    FT_CHECK_LAUNCH();
    return FT_OK;
}

template <typename K, typename PT>
static int roles_launch(K kern, size_t lds, const PT& P, hipStream_t st) {
    FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(NCU), dim3(256), lds, st, P);
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_lstm_roles_fwd)(const ft_lstm_fwd_role* roles, int n_roles, int rows_per_group, int reset_rows, void* ctx, int phase,
                                            int32_t* status, int H, void* stream) {
    FT_CHECK_ARG(roles && ctx && status && (n_roles == 1 || n_roles == 2) && reinterpret_cast<uintptr_t>(ctx) % 256 == 0 && phase >= 0);
    const int R = rows_per_group;
    FT_CHECK_ARG((R == 4 || R == 8 || R == 16) && (reset_rows == 4 || reset_rows == 8 || reset_rows == 16) && reset_rows >= 4);
    if (H != PH || !ft_lstm_persist_supported(8, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_lstm_roles_fwd: needs H == 1024 and a 256-CU device (H=%d)", H);
    FwdLaunchP P{};
    bool any = false, g16 = false;
    for (int i = 0; i < n_roles; ++i) {
        const ft_lstm_fwd_role& r = roles[i];
        FT_CHECK_ARG(r.gx && r.lens && r.y && r.wimg && r.B >= 1 && r.B <= (8 / n_roles) * R && r.ldb >= r.B && r.ldy >= H);
        FT_CHECK_ARG((r.gates == nullptr) == (r.cell == nullptr) && (r.state_h == nullptr) == (r.state_c == nullptr));
        FT_CHECK_ARG(r.t0 >= 0 && r.t1 >= r.t0 && (r.t0 == 0 || r.state_h) && reinterpret_cast<uintptr_t>(r.gx) % 16 == 0);
        P.role[i] = FwdRoleP{reinterpret_cast<const float*>(r.gx), r.lens, r.y, (long)r.ldy, r.gates, r.cell, reinterpret_cast<const unsigned short*>(r.wimg),
                             r.state_h, r.state_c, r.B, r.ldb, r.t0, r.t1};
        FT_CHECK_ARG(i == 0 || (r.gx16 != 0) == g16);             // one gx format per launch (the kernel is instantiated on it)
        g16 = r.gx16 != 0;
        any = any || r.t1 > r.t0;
    }
    if (!any) return FT_OK;
    char* base = reinterpret_cast<char*>(ctx);
    P.nroles = n_roles;
    P.census = reinterpret_cast<unsigned*>(base);
    P.hand = reinterpret_cast<unsigned*>(base + 256);
    P.phase = phase; P.reset_rows = reset_rows;
    P.status = status; P.timeout_ticks = 100000000L / 2; P.prof = ftint_roles_prof;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto lds = [](int R_, int SB_) { return sizeof(float) * ((size_t)2 * 4 * TPC * R_ * 20 + (size_t)SB_ * 4 * R_ * UPC + (size_t)2 * 6 * R_ * UPC); };
    int rc;
    const bool pf = ftint_roles_prof != nullptr;
#define FT_ROLES_FWD(R_, SB_) \
    (g16 ? (pf ? roles_launch(lstm_roles_fwd_k<R_, true, true>, lds(R_, SB_), P, st) : roles_launch(lstm_roles_fwd_k<R_, false, true>, lds(R_, SB_), P, st)) \
         : (pf ? roles_launch(lstm_roles_fwd_k<R_, true, false>, lds(R_, SB_), P, st) : roles_launch(lstm_roles_fwd_k<R_, false, false>, lds(R_, SB_), P, st)))
    if (R == 4) rc = FT_ROLES_FWD(4, 32);
    else if (R == 8) rc = FT_ROLES_FWD(8, 16);
    else rc = FT_ROLES_FWD(16, 4);
#undef FT_ROLES_FWD
    if (rc != FT_OK) return rc;
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_lstm_roles_bwd)(const ft_lstm_bwd_role* roles, int n_roles, int rows_per_group, int reset_rows, void* ctx, int phase,
                                            int32_t* status, int H, void* stream) {
    FT_CHECK_ARG(roles && ctx && status && (n_roles == 1 || n_roles == 2) && reinterpret_cast<uintptr_t>(ctx) % 256 == 0 && phase >= 0);
    const int R = rows_per_group;
    FT_CHECK_ARG((R == 4 || R == 8 || R == 16) && (reset_rows == 4 || reset_rows == 8 || reset_rows == 16));
    if (H != PH || !ft_lstm_persist_supported(8, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_lstm_roles_bwd: needs H == 1024 and a 256-CU device (H=%d)", H);
    BwdLaunchP P{};
    bool any = false;
    for (int i = 0; i < n_roles; ++i) {
        const ft_lstm_bwd_role& r = roles[i];
        FT_CHECK_ARG(r.dy && r.lens && r.gates && r.cell && (r.dgx || r.dimg) && r.wimg && r.B >= 1 && r.B <= (8 / n_roles) * R && r.ldb >= r.B && r.ldy >= H);
        FT_CHECK_ARG(r.dimg == nullptr || (r.dbias && r.ldb == r.B && r.dimg_ld >= 4 * (int64_t)H && r.dimg_ld % 8 == 0 && reinterpret_cast<uintptr_t>(r.dimg) % 16 == 0));
        FT_CHECK_ARG((r.state_da == nullptr) == (r.state_dc == nullptr) && (!r.carry_in || r.state_da) && r.t0 >= 0 && r.t1 >= r.t0);
        // (the ring is filled by 16-byte-per-lane LDS-DMA pieces: rows of the saved tensors and of dy 16-byte aligned)
        FT_CHECK_ARG(reinterpret_cast<uintptr_t>(r.dy) % 16 == 0 && r.ldy % 4 == 0 && reinterpret_cast<uintptr_t>(r.gates) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(r.cell) % 16 == 0);
        P.role[i] = BwdRoleP{r.dy, (long)r.ldy, r.lens, r.gates, r.cell, r.dgx, reinterpret_cast<const unsigned short*>(r.wimg),
                             reinterpret_cast<unsigned short*>(r.dimg), (long)r.dimg_ld, (int)r.dimg_rows, r.dbias, r.state_da, r.state_dc,
                             r.B, r.ldb, r.t0, r.t1, r.carry_in};
        any = any || r.t1 > r.t0;
    }
    if (!any) return FT_OK;
    char* base = reinterpret_cast<char*>(ctx);
    P.nroles = n_roles;
    P.census = reinterpret_cast<unsigned*>(base + 64);
    P.part = reinterpret_cast<float*>(base + 256 + 2 * FWD_SET_DW * 4);
    P.phase = phase; P.reset_rows = reset_rows;
    P.status = status; P.timeout_ticks = 100000000L / 2; P.prof = ftint_roles_prof;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto lds = [](int R_) { return sizeof(float) * ((size_t)(R_ == 4 ? 8 : 4) * UPC * R_ + 1024 + (size_t)RRING * 6 * R_ * UPC + (size_t)2 * 4 * R_ * UPC); };
    int rc;
    const bool pf = ftint_roles_prof != nullptr;
    if (R == 4) rc = pf ? roles_launch(lstm_roles_bwd_k<4, true>, lds(4), P, st) : roles_launch(lstm_roles_bwd_k<4, false>, lds(4), P, st);
    else if (R == 8) rc = pf ? roles_launch(lstm_roles_bwd_k<8, true>, lds(8), P, st) : roles_launch(lstm_roles_bwd_k<8, false>, lds(8), P, st);
    else rc = pf ? roles_launch(lstm_roles_bwd_k<16, true>, lds(16), P, st) : roles_launch(lstm_roles_bwd_k<16, false>, lds(16), P, st);
    if (rc != FT_OK) return rc;
    FT_CHECK_LAUNCH();
    return FT_OK;
}
