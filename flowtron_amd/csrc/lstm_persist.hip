// Persistent LSTM recurrence (forward): ONE launch for the whole sequence, weights resident in the register file.
//
// The launch-per-step kernels of lstm.hip re-stream W_hh (8.4 MB bf16 at H = 1024) from the memory side every time step
// because nothing survives a kernel boundary in registers/LDS, and every CU pulls the whole [B,H] state image: ~5.3 us per
// step, 2.5x the ~2 us a dependent step needs.  Here the chip is split into NG independent BATCH GROUPS:
//
//   group g  = 256/NG workgroups (one per CU, 4 waves, one wave per SIMD = the whole 512-register file per wave)
//              owns batch rows [g*RPG, (g+1)*RPG) for the WHOLE sequence -- no traffic between groups, ever;
//   CU q of a group owns UPC = H*NG/256 hidden units = TPC = NG tiles of (4 units x 4 gates) gate rows and keeps
//              their W_hh rows as bf16 MFMA fragments in registers (wave w: k-chunks w, w+4, ..: 32*NG VGPRs);
//   per step   each wave reads its quarter of the group's h_{t-1} (RPG rows x H bf16, <= 8 KB per group) straight into
//              MFMA A-fragments, TPC x 8 MFMAs per wave, 4-wave reduce in LDS, cell update for RPG x UPC (= 128) elements,
//              h_t published to the group.
//
// W_hh is replicated NG times across the chip (registers are plentiful: 512 KB per CU), which shrinks the per-step all-gather
// from the full [32,1024] state to [RPG,1024] and makes it group-local.
//
// Hand-off (cdna_hip_programming.md G16, form R2 "the data IS the flag"): h_t travels as 8-byte granules
// {hi = epoch t+1, lo = two bf16}, one aligned 8-byte store each; consumers re-read their granules with L1-bypassing
// 16-byte loads (two granules) until every tag matches -- no fences, no separate flag.  Two parity buffers: a producer can
// only overwrite slot parity p two steps later, after every consumer of the group has published the step in between, i.e.
// has finished reading p.  Two transports (template LOCAL):
//   LOCAL = false  placement independent: write-through (sc1) stores, sc1 loads -- the data crosses the fabric each step
//                  (~2 us per hop measured); groups are formed from block ids; NG = 8 | 4 | 2.
//   LOCAL = true   NG = 8, group == XCD: every workgroup reads its XCC id and draws its slot in that XCD's group from a
//                  per-XCD counter (a census: membership is a FACT established at run time, not a dispatch-order
//                  assumption; with one 512-register workgroup per CU each XCD hosts exactly 32).  Producers then use
//                  PLAIN stores -- the CU's L1 is write-through, so the granule sits in the XCD's own L2 -- and consumers
//                  sc1 loads, which bypass L1 and are served by that same L2: one L2 round trip per hop instead of the fabric.
//                  A stale tag can never be mistaken for data; if the census does not come out (partitioned device, foreign
//                  kernel on some CUs) the waits time out and the status word sends the caller to the launch-per-step path.
// Every spin is bounded by a wall-clock timeout that raises status[0] (the host checks it; a chip with fewer than 256 free
// CUs cannot host the grid).
//
// Saved tensors (gates, cell), y, masks: identical to ft_lstm_seq_fwd (FT_BF16 path) -- same fragment rounding, same
// k-chunk-per-wave accumulation order, so results are bit-identical to the launch-per-step kernel.
#include "common.h"
#include "lstm_images.h"

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int PH = 1024;                 // hidden size this kernel is built for
constexpr int NCHUNK = PH / 32;          // 32 k-chunks of 32
constexpr int NCU = 256;

struct PersistP {
    const float* gx; const int* lens;
    float* y; long ldy; float* gates; float* cell;
    const unsigned short* wfrag;         // [H/4][H/32][64][8] bf16 (make_wfrag_fwd layout)
    unsigned long long* hgran;           // [2 parity][NG][NCHUNK][4 kg][RPGP][4] granules
    int* status;
    unsigned* census;                    // LOCAL: [8] per-XCD arrival counters (zeroed by the host before the launch)
    int T, B;
    long timeout_ticks;                  // wall_clock64 ticks (100 MHz)
};

// granule index of (b, k) inside one group's buffer: [c = k>>5][kg = (k>>3)&3][b][pair = (k&7)>>1]
template <int RPGP>
__device__ __forceinline__ int gran_index(int b, int k) {
    return ((((k >> 5) * 4 + ((k >> 3) & 3)) * RPGP + b) << 2) + ((k & 7) >> 1);
}

template <int NG, bool LOCAL>
__global__ __launch_bounds__(256, 1) void lstm_persist_fwd_k(PersistP p) {
    static_assert(!LOCAL || NG == 8, "the L2-local transport needs group == XCD");
    constexpr int CPG = NCU / NG;                  // CUs (workgroups) per group
    constexpr int UPC = PH / CPG;                  // hidden units per CU: 32, 16, 8
    constexpr int TPC = UPC / 4;                   // gate-row tiles per CU (= NG)
    constexpr int RPGP = 32 / NG;                  // batch rows per group (padded): 4, 8, 16
    constexpr int GRAN_PER_GROUP = NCHUNK * 4 * RPGP * 4;
    __shared__ float red[2][4][TPC][RPGP][17];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int grp, q;
    if constexpr (LOCAL) {
        __shared__ int slot[2];
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            slot[0] = (int)(xcc & 7u);
            slot[1] = (int)__hip_atomic_fetch_add(p.census + (xcc & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        grp = __builtin_amdgcn_readfirstlane(slot[0]); q = __builtin_amdgcn_readfirstlane(slot[1]);
        if (q >= CPG) {                                          // more than 32 workgroups on one XCD: not the machine this is for
            if (tid == 0) atomicExch(p.status, 2);
            return;
        }
    } else {
        grp = blockIdx.x % NG; q = blockIdx.x / NG;              // speed only: consecutive block ids land on different XCDs
    }
    const int B = p.B, T = p.T;
    const int b0 = grp * RPGP;

    // ---- resident weights: tile j, k-chunk (wave + 4 i)
    bf16x8 w[TPC][8];
    {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.wfrag);
#pragma unroll
        for (int j = 0; j < TPC; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) w[j][i] = wf[((size_t)(q * TPC + j) * NCHUNK + (wave + 4 * i)) * 64 + lane];
    }

    // ---- epilogue role: thread e < 128 owns (batch row eb, unit eu) of this CU for the whole sequence
    const bool erole = tid < RPGP * UPC;           // == 128
    const int el = tid % UPC, ebl = tid / UPC;     // unit within CU, row within group
    const int eb = b0 + ebl, eu = q * UPC + el;
    const bool ev = erole && eb < B;
    const int ebc = eb < B ? eb : B - 1;
    const int len = erole ? p.lens[ebc] : 0;
    // steps this group runs: the longest sequence among its rows (uniform per workgroup)
    int tg = 0;
#pragma unroll
    for (int r = 0; r < RPGP; ++r) {
        const int bb = b0 + r;
        if (bb < B) { const int l = p.lens[bb]; tg = l > tg ? l : tg; }
    }
    tg = tg < T ? tg : T;

    float c_state = 0.f, h_state = 0.f;
    float gxv[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_gx = [&](int t) {
        if (erole && t < T) {
            const float* gp = p.gx + ((size_t)t * B + ebc) * 4 * PH + eu;
#pragma unroll
            for (int g = 0; g < 4; ++g) gxv[g] = gp[(size_t)g * PH];
        }
    };
    load_gx(0);

    __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        rs[par] = __builtin_amdgcn_make_buffer_rsrc(p.hgran + ((size_t)par * NG + grp) * GRAN_PER_GROUP, 0,
                                                    GRAN_PER_GROUP * 8, 0x00020000);
    // per-lane byte offset of chunk 0 (chunk c adds c * 4 * RPGP * 32 bytes): [kg][b = li % RPGP][4 granules]
    const int voff0 = ((kg * RPGP + (li % RPGP)) * 4) * 8, voff1 = voff0 + 16;
    // chunk offsets ride in the scalar offset operand (wave-uniform), so the 16 loads share two address VGPRs
    const int soff_w = __builtin_amdgcn_readfirstlane(wave) * (4 * RPGP * 32);
    const long t_start = wall_clock64();
    bool dead = false;

    for (int t = 0; t < tg; ++t) {
        f32x4 acc[TPC];
#pragma unroll
        for (int j = 0; j < TPC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            // ---- sweep: this wave's 8 chunks of h_{t-1} (epoch t) straight into A fragments.  Every pass re-reads ALL chunks
            // that have not shown the epoch yet (one round trip for the lot), and the MFMAs run in chunk ORDER (deterministic
            // accumulation) over the ready prefix, so early chunks are multiplied while later producers still publish.
            const unsigned epoch = (unsigned)t;
            const int par = (t - 1) & 1;
            u32x4 lo[8], hi[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int so = soff_w + 4 * i * (4 * RPGP * 32);
                lo[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff0, so, 16);       // aux 16 = sc1: bypass L1
                hi[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff1, so, 16);
            }
            unsigned ready = 0;                    // wave-uniform bit per chunk: tags matched
            int next = 0;                          // chunks [0, next) are already multiplied
            for (unsigned spins = 0;; ++spins) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (!((ready >> i) & 1u)) {
                        const bool ok = (lo[i][1] == epoch) & (lo[i][3] == epoch) & (hi[i][1] == epoch) & (hi[i][3] == epoch);
                        if (__all(ok)) ready |= 1u << i;
                    }
                    if (next == i && ((ready >> i) & 1u)) {
                        const u32x4 v = {lo[i][0], lo[i][2], hi[i][0], hi[i][2]};
                        const bf16x8 a = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                        for (int j = 0; j < TPC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w[j][i], acc[j], 0, 0, 0);
                        next = i + 1;
                    }
                }
                if (next == 8) break;
                if ((spins & 15) == 15) {
                    if (wall_clock64() - t_start > p.timeout_ticks ||
                        __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        dead = true;
                        break;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (!((ready >> i) & 1u)) {
                        const int so = soff_w + 4 * i * (4 * RPGP * 32);
                        lo[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff0, so, 16);
                        hi[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff1, so, 16);
                    }
                }
            }
            if (dead) break;
        }
        // D[m = batch row (lane>>4)*4 + r][n = li]: rows >= RPGP are padding
        const int rb = t & 1;
        if (kg * 4 < RPGP) {
#pragma unroll
            for (int j = 0; j < TPC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[rb][wave][j][kg * 4 + r][li] = acc[j][r];
        }
        __syncthreads();
        if (erole) {
            const bool active = t < len;
            const int j = el >> 2, ul = el & 3;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = g * 4 + ul;
                pre[g] = red[rb][0][j][ebl][n] + red[rb][1][j][ebl][n] + red[rb][2][j][ebl][n] + red[rb][3][j][ebl][n] + gxv[g];
            }
            float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f;
            if (active) {
                float c_new, h_new;
                lstm_cell<true>(pre, c_state, ig, fg, gg, og, c_new, h_new);
                c_state = c_new; h_state = h_new;
            }
            // ---- publish h_t first: one 8-byte {epoch, bf16 pair} granule per even unit (frozen rows re-publish their state)
            const float h_nb = __shfl_down(h_state, 1, 64);
            if ((el & 1) == 0) {
                const unsigned long long gran = ((unsigned long long)(unsigned)(t + 1) << 32) | pack_bf16x2(h_state, h_nb);
                unsigned long long* dst = p.hgran + ((size_t)(t & 1) * NG + grp) * GRAN_PER_GROUP + gran_index<RPGP>(ebl, eu);
                // LOCAL: workgroup-scope relaxed store = ONE aligned 8-byte global_store (sc0) whose line stays in this XCD's L2;
                // otherwise agent scope = sc1, write-through to the memory side
                if constexpr (LOCAL) __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (ev) {
                const size_t row = (size_t)t * B + eb;
                p.y[row * p.ldy + eu] = active ? h_state : 0.f;
                if (active && p.gates) {
                    float* gp = p.gates + row * 4 * PH + eu;
                    gp[0] = ig; gp[(size_t)PH] = fg; gp[(size_t)2 * PH] = gg; gp[(size_t)3 * PH] = og;
                    p.cell[row * PH + eu] = c_state;
                }
            }
            load_gx(t + 1);
        }
    }
    if (dead) {
        if (lane == 0) atomicExch(p.status, 1);
        return;
    }
    // pad rows beyond the group's longest sequence: y = 0 (pad_packed_sequence semantics)
    if (ev)
        for (int t = tg; t < T; ++t) p.y[((size_t)t * B + eb) * p.ldy + eu] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward recurrence, same organisation: dh_rec[b][j] = sum_r dgates_{s+1}[b][r] W_hh[r][j]  (K = 4H, N = H).
// CU q of a group owns UPC hidden units j (UPC/16 column tiles of W_hh^T, fragments resident in registers: wave w holds
// k-chunks w, w+4, .. of K = 4H = 128 chunks) and runs the cell backward of those units for the group's rows, publishing
// dgates_s (4 gates x UPC units x RPGP rows) as granules over k = gate*H + j.  The group's dgates vector is 4x the forward
// state (RPGP x 4H bf16), so a wave sweeps its 32 chunks in 4 batches of 8 with the next batch's loads in flight.
// Accumulation mimics lstm_bwd_step_bf16's 16-wave split (partial a of wave w = chunks w+4a, w+4a+16, ..; the 16 partials
// are summed in wave order), so the result is bit-identical to the launch-per-step kernel.
struct PersistBwdP {
    const float* dy; long ldy; const int* lens;
    const float* gates; const float* cell; float* dgx;
    const unsigned short* wTfrag;        // [H/16][4H/32][64][8] bf16 (make_wfrag_bwd layout)
    unsigned long long* dgran;           // [2 parity][NG][4H/32][4 kg][RPGP][4] granules
    int* status; unsigned* census;
    int T, B;
    long timeout_ticks;
};

template <int NG, bool LOCAL>
__global__ __launch_bounds__(256, 1) void lstm_persist_bwd_k(PersistBwdP p) {
    static_assert(!LOCAL || NG == 8, "the L2-local transport needs group == XCD");
    constexpr int CPG = NCU / NG;                  // workgroups per group
    constexpr int UPC = PH / CPG;                  // hidden units per CU: 32 (NG 8), 16 (NG 4)
    constexpr int TL = UPC / 16;                   // column tiles per CU
    constexpr int RPGP = 32 / NG;
    constexpr int KCH = 4 * PH / 32;               // 128 k-chunks
    constexpr int CPW = KCH / 4;                   // chunks per wave: 32
    constexpr int GRAN_PER_GROUP = KCH * 4 * RPGP * 4;
    static_assert(TL >= 1, "NG = 2 would leave half a column tile per CU");
    __shared__ float red[2][16][TL][RPGP][17];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int grp, q;
    if constexpr (LOCAL) {
        __shared__ int slot[2];
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            slot[0] = (int)(xcc & 7u);
            slot[1] = (int)__hip_atomic_fetch_add(p.census + (xcc & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        grp = __builtin_amdgcn_readfirstlane(slot[0]); q = __builtin_amdgcn_readfirstlane(slot[1]);
        if (q >= CPG) {
            if (tid == 0) atomicExch(p.status, 2);
            return;
        }
    } else {
        grp = blockIdx.x % NG; q = blockIdx.x / NG;
    }
    const int B = p.B, T = p.T;
    const int b0 = grp * RPGP;

    bf16x8 w[TL][CPW];
    {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.wTfrag);
#pragma unroll
        for (int j = 0; j < TL; ++j)
#pragma unroll
            for (int i = 0; i < CPW; ++i) w[j][i] = wf[((size_t)(q * TL + j) * KCH + (wave + 4 * i)) * 64 + lane];
    }

    const bool erole = tid < RPGP * UPC;           // == 128
    const int el = tid % UPC, ebl = tid / UPC;
    const int eb = b0 + ebl, eu = q * UPC + el;
    const bool ev = erole && eb < B;
    const int ebc = eb < B ? eb : B - 1;
    const int len = erole ? p.lens[ebc] : 0;
    int tg = 0;
#pragma unroll
    for (int r = 0; r < RPGP; ++r) {
        const int bb = b0 + r;
        if (bb < B) { const int l = p.lens[bb]; tg = l > tg ? l : tg; }
    }
    tg = tg < T ? tg : T;

    // saved activations of the step being processed, and of the next one (one step of prefetch from HBM)
    float g4[4] = {0.f, 0.f, 0.f, 0.f}, c_t = 0.f, c_prev = 0.f, dyv = 0.f;
    float ng4[4] = {0.f, 0.f, 0.f, 0.f}, ncp = 0.f, ndy = 0.f;
    auto load_step = [&](int s, float (&gg_)[4], float& cprev_, float& dy_) {     // gates[s], cell[s-1], dy[s]
        if (erole && s >= 0) {
            const size_t row = (size_t)s * B + ebc;
            const float* gp = p.gates + row * 4 * PH + eu;
#pragma unroll
            for (int g = 0; g < 4; ++g) gg_[g] = gp[(size_t)g * PH];
            cprev_ = s > 0 ? p.cell[((size_t)(s - 1) * B + ebc) * PH + eu] : 0.f;
            dy_ = p.dy[row * p.ldy + eu];
        }
    };
    if (erole && tg > 0) c_t = p.cell[((size_t)(tg - 1) * B + ebc) * PH + eu];
    load_step(tg - 1, g4, c_prev, dyv);
    load_step(tg - 2, ng4, ncp, ndy);
    float dc_carry = 0.f;

    __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        rs[par] = __builtin_amdgcn_make_buffer_rsrc(p.dgran + ((size_t)par * NG + grp) * GRAN_PER_GROUP, 0,
                                                    GRAN_PER_GROUP * 8, 0x00020000);
    const int voff0 = ((kg * RPGP + (li % RPGP)) * 4) * 8, voff1 = voff0 + 16;
    constexpr int CHUNK_BYTES = 4 * RPGP * 32;
    const int soff_w = __builtin_amdgcn_readfirstlane(wave) * CHUNK_BYTES;
    const long t_start = wall_clock64();
    bool dead = false;

    for (int n = 0; n < tg; ++n) {                 // n-th step of the sweep: time index s = tg-1-n
        const int s = tg - 1 - n;
        f32x4 acc[TL][4];
#pragma unroll
        for (int j = 0; j < TL; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[j][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (n > 0) {
            const unsigned epoch = (unsigned)n;
            const int par = (n - 1) & 1;
            u32x4 lo[2][8], hi[2][8];
            auto issue = [&](int bt, u32x4 (&l_)[8], u32x4 (&h_)[8]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int so = soff_w + 4 * (bt * 8 + i) * CHUNK_BYTES;
                    l_[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff0, so, 16);
                    h_[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff1, so, 16);
                }
            };
            issue(0, lo[0], hi[0]);
#pragma unroll
            for (int bt = 0; bt < 4; ++bt) {
                if (bt < 3) issue(bt + 1, lo[(bt + 1) & 1], hi[(bt + 1) & 1]);
                u32x4 (&L_)[8] = lo[bt & 1];
                u32x4 (&H_)[8] = hi[bt & 1];
                unsigned ready = 0;
                int next = 0;
                for (unsigned spins = 0;; ++spins) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (!((ready >> i) & 1u)) {
                            const bool ok = (L_[i][1] == epoch) & (L_[i][3] == epoch) & (H_[i][1] == epoch) & (H_[i][3] == epoch);
                            if (__all(ok)) ready |= 1u << i;
                        }
                        if (next == i && ((ready >> i) & 1u)) {
                            const int ci = bt * 8 + i;               // this wave's ci-th chunk = global chunk wave + 4 ci
                            const u32x4 v = {L_[i][0], L_[i][2], H_[i][0], H_[i][2]};
                            const bf16x8 a = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                            for (int j = 0; j < TL; ++j)
                                acc[j][ci & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w[j][ci], acc[j][ci & 3], 0, 0, 0);
                            next = i + 1;
                        }
                    }
                    if (next == 8) break;
                    if ((spins & 15) == 15) {
                        if (wall_clock64() - t_start > p.timeout_ticks ||
                            __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            dead = true;
                            break;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (!((ready >> i) & 1u)) {
                            const int so = soff_w + 4 * (bt * 8 + i) * CHUNK_BYTES;
                            L_[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff0, so, 16);
                            H_[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff1, so, 16);
                        }
                    }
                }
                if (dead) break;
            }
            if (dead) break;
        }
        const int rb = n & 1;
        if (kg * 4 < RPGP) {
#pragma unroll
            for (int j = 0; j < TL; ++j)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[rb][wave + 4 * a][j][kg * 4 + r][li] = acc[j][a][r];
        }
        __syncthreads();
        if (erole) {
            const bool active = s < len;
            float da[4] = {0.f, 0.f, 0.f, 0.f};
            if (active) {
                const int j = el >> 4, nn = el & 15;
                float dh = dyv;
#pragma unroll
                for (int w16 = 0; w16 < 16; ++w16) dh += red[rb][w16][j][ebl][nn];
                float carry;
                lstm_cell_bwd<true>(dh, dc_carry, g4[0], g4[1], g4[2], g4[3], c_t, c_prev, da, carry);
                dc_carry = carry;
            }
            // ---- publish dgates_s: one granule per gate per even unit (k = gate*H + unit)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float nb = __shfl_down(da[g], 1, 64);
                if ((el & 1) == 0) {
                    const unsigned long long gran = ((unsigned long long)(unsigned)(n + 1) << 32) | pack_bf16x2(da[g], nb);
                    unsigned long long* dst = p.dgran + ((size_t)(n & 1) * NG + grp) * GRAN_PER_GROUP + gran_index<RPGP>(ebl, g * PH + eu);
                    if constexpr (LOCAL) __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (ev) {
                float* dg = p.dgx + ((size_t)s * B + eb) * 4 * PH + eu;
#pragma unroll
                for (int g = 0; g < 4; ++g) dg[(size_t)g * PH] = da[g];
            }
            // rotate the prefetched step in and fetch the one after
#pragma unroll
            for (int g = 0; g < 4; ++g) g4[g] = ng4[g];
            c_t = c_prev; c_prev = ncp; dyv = ndy;      // cell[s-1] is c_prev of step s and c_t of step s-1
            load_step(s - 2, ng4, ncp, ndy);
        }
    }
    if (dead) {
        if (lane == 0) atomicExch(p.status, 1);
        return;
    }
    if (ev)                                                             // pad rows beyond the group's longest sequence
        for (int t = tg; t < T; ++t) {
            float* dg = p.dgx + ((size_t)t * B + eb) * 4 * PH + eu;
#pragma unroll
            for (int g = 0; g < 4; ++g) dg[(size_t)g * PH] = 0.f;
        }
}

}  // namespace

static inline size_t al256p(size_t v) { return (v + 255) & ~size_t(255); }

extern "C" int ft_lstm_persist_supported(int B, int H) {
    if (H != PH || B < 1 || B > 32) return 0;
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    return cus >= NCU ? 1 : 0;
}

extern "C" size_t ft_lstm_persist_workspace_bytes(int B, int H) {
    (void)B;
    // W_hh fragment image + granule buffers (2 parities x 32 rows x K/2 granules x 8 B, independent of NG; K = H forward,
    // 4H backward) + census counters
    return al256p((size_t)4 * H * H * 2) + al256p((size_t)2 * 32 * (4 * H / 2) * 8) + 256;
}

extern "C" int ft_lstm_persist_fwd(const float* gx, const float* w_hh, const int32_t* lens, float* y, int64_t ldy,
                                   float* gates, float* cell, void* work, int32_t* status, int T, int B, int H, int ng,
                                   void* stream) {
    FT_CHECK_ARG(gx && w_hh && lens && y && work && status);
    FT_CHECK_ARG((gates == nullptr) == (cell == nullptr));
    FT_CHECK_ARG(T >= 0 && ldy >= H && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    FT_CHECK_ARG(ng == 1 || ng == 8 || ng == 4 || ng == 2);       // 1 = XCD-local transport (8 groups = 8 XCDs)
    if (!ft_lstm_persist_supported(B, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_lstm_persist_fwd: needs H == 1024, B <= 32 and a 256-CU device (H=%d B=%d)", H, B);
    if (T == 0) return FT_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(work);
    unsigned short* wfrag = reinterpret_cast<unsigned short*>(base);
    unsigned long long* hgran = reinterpret_cast<unsigned long long*>(base + al256p((size_t)4 * H * H * 2));
    const size_t gran_bytes = al256p((size_t)2 * 32 * (H / 2) * 8);
    unsigned* census = reinterpret_cast<unsigned*>(base + al256p((size_t)4 * H * H * 2) + al256p((size_t)2 * 32 * (4 * H / 2) * 8));
    FT_CHECK_HIP(hipMemsetAsync(hgran, 0, gran_bytes, st));           // tags = 0: no epoch matches (epochs start at 1)
    FT_CHECK_HIP(hipMemsetAsync(census, 0, 256, st));
    hipLaunchKernelGGL(make_wfrag_fwd, dim3(2048), dim3(256), 0, st, w_hh, wfrag, H);
    PersistP p{gx, lens, y, (long)ldy, gates, cell, wfrag, hgran, status, census, T, B, 100000000L / 2};   // 0.5 s
    if (ng == 1) hipLaunchKernelGGL((lstm_persist_fwd_k<8, true>), dim3(NCU), dim3(256), 0, st, p);
    else if (ng == 8) hipLaunchKernelGGL((lstm_persist_fwd_k<8, false>), dim3(NCU), dim3(256), 0, st, p);
    else if (ng == 4) hipLaunchKernelGGL((lstm_persist_fwd_k<4, false>), dim3(NCU), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lstm_persist_fwd_k<2, false>), dim3(NCU), dim3(256), 0, st, p);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int ft_lstm_persist_bwd(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                                   const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng,
                                   void* stream) {
    FT_CHECK_ARG(dy && w_hh && lens && gates && cell && dgx && work && status);
    FT_CHECK_ARG(T >= 0 && ldy >= H && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    FT_CHECK_ARG(ng == 1 || ng == 8 || ng == 4);
    if (!ft_lstm_persist_supported(B, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_lstm_persist_bwd: needs H == 1024, B <= 32 and a 256-CU device (H=%d B=%d)", H, B);
    if (T == 0) return FT_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(work);
    unsigned short* wTfrag = reinterpret_cast<unsigned short*>(base);
    unsigned long long* dgran = reinterpret_cast<unsigned long long*>(base + al256p((size_t)4 * H * H * 2));
    const size_t gran_bytes = al256p((size_t)2 * 32 * (4 * H / 2) * 8);
    unsigned* census = reinterpret_cast<unsigned*>(base + al256p((size_t)4 * H * H * 2) + gran_bytes);
    FT_CHECK_HIP(hipMemsetAsync(dgran, 0, gran_bytes + 256, st));
    hipLaunchKernelGGL(make_wfrag_bwd, dim3(2048), dim3(256), 0, st, w_hh, wTfrag, H);
    PersistBwdP p{dy, (long)ldy, lens, gates, cell, dgx, wTfrag, dgran, status, census, T, B, 100000000L / 2};
    if (ng == 1) hipLaunchKernelGGL((lstm_persist_bwd_k<8, true>), dim3(NCU), dim3(256), 0, st, p);
    else if (ng == 8) hipLaunchKernelGGL((lstm_persist_bwd_k<8, false>), dim3(NCU), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lstm_persist_bwd_k<4, false>), dim3(NCU), dim3(256), 0, st, p);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
