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
#include "lstm_persist_common.h"

namespace {

struct PersistP {
    const float* gx; const int* lens;
    float* y; long ldy; float* gates; float* cell;
    const unsigned short* wfrag;         // [H/4][H/32][64][8] bf16 (make_wfrag_fwd layout)
    unsigned long long* hgran;           // [2 parity][NG][NCHUNK][4 kg][RPGP][4] granules
    int* status;
    unsigned* census;                    // LOCAL: [8] per-XCD arrival counters (zeroed by the host before the launch)
    int T, B;
    long timeout_ticks;                  // wall_clock64 ticks (100 MHz)
    long* prof;                          // debug: per-step phase stamps of one workgroup (ft_lstm_persist_debug_prof), or null
    int LB;                              // batch rows per time step IN MEMORY (>= B: the launch may cover a slice b0 .. b0 + B - 1 of a
                                         // wider batch -- every pointer then starts at row b0; ft_lstm_persist_fwd_rows)
};

// Granule layout of one group's state vector (K = 32 NCW k-values x RPGP rows; NCW = k-chunks per wave).  A consumer wave w
// owns the chunks c = w + 4 ci; ONE 16-byte-per-lane load fetches CPL = 16 / RPGP of its chunks at once -- lane (kg, li) gets
// chunk ci = lg CPL + li / RPGP, row li % RPGP, k-group kg -- so every lane of every load carries real granules (a load per
// chunk would leave the lanes of the padding rows, 3/4 of the wave at RPGP = 4, fetching duplicates).  MFMA wants chunk j's rows
// in lanes li < RPGP of each 16-lane row: a DPP row shift by j RPGP lanes puts them there (the other lanes are padding rows
// whose products nobody reads).  Buffer: [w][lg][half][64 lanes][2 granules]; half = which 4 of the lane's 8 k-values.
template <int RPGP, int NCW>
__device__ __forceinline__ int gran_index(int b, int k) {
    constexpr int CPL = 16 / RPGP, NLG = NCW / CPL;
    const int c = k >> 5, w = c & 3, ci = c >> 2, kg = (k >> 3) & 3, e = k & 7;
    const int lg = ci / CPL, j = ci % CPL, lane = kg * 16 + j * RPGP + b;
    return ((((w * NLG + lg) * 2 + (e >> 2)) * 64 + lane) << 1) + ((e >> 1) & 1);
}

// the eight operands of one lane from the load(s) of a load group: tagged = two loads {v, tag, v, tag}, bare = one load
template <bool BARE>
__device__ __forceinline__ u32x4 payload(const u32x4* l) {
    if constexpr (BARE) return l[0];
    else return (u32x4){l[0][0], l[0][2], l[1][0], l[1][2]};
}
template <bool BARE>
__device__ __forceinline__ bool fresh(const u32x4* l, unsigned epoch) {
    if constexpr (BARE) return (l[0][0] != SENT) & (l[0][1] != SENT) & (l[0][2] != SENT) & (l[0][3] != SENT);
    else return (l[0][1] == epoch) & (l[0][3] == epoch) & (l[1][1] == epoch) & (l[1][3] == epoch);
}
// Staging of the HBM rows the recurrence reads.
// Forward: gx rows arrive in synchronous bursts of SB steps (one HBM round trip per 32 steps, ~0.1 us per step).
// Backward (saved gates, cell, dy: 6 rows per step, 97 KB per burst -- measured 20 us of whole-workgroup stall per 32 steps,
// 0.6 us per step): a RING of LDS slots, one step per slot, filled DIST steps ahead by LDS-DMA from waves 2-3 -- six dwords per
// lane and step, issued right after the reduce barrier together with the output stores of the previous step.  Those waves have
// ~1 us of slack per step (they wait for the group's publish), which hides the HBM round trip that, vmcnt retiring in order,
// sits in front of their next poll; the forward kernel's waves have no such slack (tried: 1.93 -> 2.15 us per step), and a DMA
// issued BEFORE the barrier stalls everybody (LDS-DMA also counts in lgkmcnt, which every LDS barrier has to drain).
// Slot reuse: the occupant of slot m % RING (step m - RING) is last read in step m - RING + 1 (cell ring); its successor is
// requested in step m - DIST > m - RING + 1.
constexpr int SB = 32;
constexpr int RING = 8, DIST = 6;
static_assert(DIST < RING - 1, "slot reuse");

template <int NG, bool LOCAL, int LAUX, bool BARE>
__global__ __launch_bounds__(256, 1) void lstm_persist_fwd_k(PersistP p) {
    static_assert(!LOCAL || NG == 8, "the L2-local transport needs group == XCD");
    constexpr int CPG = NCU / NG;                  // CUs (workgroups) per group
    constexpr int UPC = PH / CPG;                  // hidden units per CU: 32, 16, 8
    constexpr int TPC = UPC / 4;                   // gate-row tiles per CU (= NG)
    constexpr int RPGP = 32 / NG;                  // batch rows per group (padded): 4, 8, 16
    constexpr int NE = RPGP * UPC;                 // (row, unit) elements per CU = 128 epilogue threads
    constexpr int CPL = 16 / RPGP, NLG = 8 / CPL;  // chunks per load, load groups per wave (x 2 halves)
    constexpr int GRAN_PER_GROUP = NCHUNK * 4 * RPGP * 4;
    static_assert(NE == 128, "one epilogue element per thread of waves 0-1");
    // LDS: 4-wave reduce (double buffered) | SB steps of gx rows [s][gate][e] | 2 steps of outputs [parity][y,i,f,g,o,c][e]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (rows of 16 gate columns at a 20-float pitch: 16-byte aligned for the epilogue's f32x4 reads, column n = unit * 4 + gate)
    float (*red)[4][TPC][RPGP][20] = reinterpret_cast<float (*)[4][TPC][RPGP][20]>(smem);
    float* gxs = smem + 2 * 4 * TPC * RPGP * 20;
    float* outs = gxs + SB * 4 * NE;


    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int grp, q;
    if constexpr (LOCAL) {
        if (!join_group_local<CPG>(p.census, p.status, tid, grp, q)) return;
    } else {
        grp = blockIdx.x % NG; q = blockIdx.x / NG;              // speed only: consecutive block ids land on different XCDs
    }
    const int B = p.B, T = p.T, LB = p.LB;
    const int b0 = grp * RPGP;

    // ---- resident weights: tile j, k-chunk (wave + 4 i).  NG == 8 (64 fragments = 256 registers per lane): parked in ACCUMULATION
    // registers for the whole launch and fed to the MFMAs from there (mfma16_bagpr: round 4 -- the builtin takes B from VGPRs only, so
    // the ~17 fragments the allocator had to keep in AGPRs cost four v_accvgpr_read each on every use, issue slots the step's MFMA
    // block does not hide); accumulators in VGPRs.  Same MFMA order per accumulator: bit-identical to the launch-per-step kernel.
    constexpr bool WAGPR = NG == 8;
    bf16x8 w[TPC][8];
    {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.wfrag);
#pragma unroll
        for (int j = 0; j < TPC; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) w[j][i] = wf[((size_t)(q * TPC + j) * NCHUNK + (wave + 4 * i)) * 64 + lane];
        if constexpr (WAGPR) {
#pragma unroll
            for (int j = 0; j < TPC; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(w[j][i]));
        }
    }

    // ---- epilogue role: thread e < 128 owns (batch row eb, unit eu) of this CU for the whole sequence
    const bool erole = tid < NE;
    const int el = tid % UPC, ebl = tid / UPC;     // unit within CU, row within group
    const int eb = b0 + ebl, eu = q * UPC + el;
    const bool ev = erole && eb < B;
    const int len = ev ? p.lens[eb] : 0;
    // steps this group runs: the longest sequence among its rows (uniform per workgroup)
    int tg = 0;
#pragma unroll
    for (int r = 0; r < RPGP; ++r) {
        const int bb = b0 + r;
        if (bb < B) { const int l = p.lens[bb]; tg = l > tg ? l : tg; }
    }
    tg = tg < T ? tg : T;

    // ---- HBM traffic is kept out of the sweeping waves' way.  hipcc waits vmcnt(0) at the block joins of the poll loop, and
    // vmcnt retires in order and counts stores, so anything in flight when a sweep starts sits in front of its first wait:
    //   * gx rows come in bursts of SB steps by LDS-DMA (global_load_lds: no VGPR staging), one HBM round trip per SB steps;
    //   * the saved tensors of step t-1 are stored by waves 2-3 right after the reduce barrier of step t, while waves 0-1 run
    //     the cell update: waves 2-3 start polling before the group has published, so their first (failing) pass hides the
    //     store acknowledgements.
    const int wu = __builtin_amdgcn_readfirstlane(wave);         // wave id as a scalar
    const int eh = (wave & 1) * 64 + lane;                       // burst: wave w serves elements 64 (w & 1) + lane
    const int hb = b0 + eh / UPC, hu = q * UPC + eh % UPC;
    const bool hvalid = hb < B;
    // (RPGP == 4: 1 KiB pieces -- global_load_lds_dwordx4, 16 bytes per lane, eight lanes per 128-byte row piece: one piece = two gates of
    //  one step for the group's four rows; 16 pieces per wave and burst instead of 64 dword pieces: a piece costs 60-185 cycles of issue
    //  time whatever its size.  Piece c = 4 kk + wave: step c >> 1, gates 2 (c & 1) + (lane >> 5); lane -> row (lane & 31) >> 3, units 4 (lane & 7))
    const int pr = lane >> 5, pb = (lane & 31) >> 3, pu = 4 * (lane & 7);
    const bool pvalid = b0 + pb < B;
    auto burst = [&](int t) {                                    // t % SB == 0: gx rows of steps [t, t + SB)
        __syncthreads();
        const int nst = (tg - t) < SB ? (tg - t) : SB;
        if constexpr (RPGP == 4) {
            if (pvalid) {
                const float* src0 = p.gx + ((size_t)t * LB + b0 + pb) * 4 * PH + (size_t)pr * PH + q * UPC + pu;
                const unsigned dst0 = (unsigned)(size_t)(lds_void*)gxs;
#pragma unroll
                for (int kk = 0; kk < SB / 2; ++kk) {
                    const int c = 4 * kk + wu, st = c >> 1, h = c & 1;
                    if (st < nst) {
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(src0 + (size_t)st * LB * 4 * PH + (size_t)(2 * h) * PH),
                                       "s"(dst0 + (unsigned)((st * 4 + 2 * h) * NE * 4)) : "memory");
                    }
                }
            }
        } else if (hvalid) {
            const float* src0 = p.gx + ((size_t)t * LB + hb) * 4 * PH + hu;
            const unsigned dst0 = (unsigned)(size_t)(lds_void*)gxs + (unsigned)(wu & 1) * 256u;
#pragma unroll
            for (int kk = 0; kk < SB * 2; ++kk) {                // pair k = (step, gate), LDS slot [k][e]; all DMAs in flight
                const int k = 2 * kk + (wu >> 1);
                if (k < nst * 4) dma_dword(src0 + (size_t)(k >> 2) * LB * 4 * PH + (size_t)(k & 3) * PH, dst0 + (unsigned)k * NE * 4u);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the DMA writes are this wave's own VM operations
        __syncthreads();
    };
    // output role of waves 2-3: thread tid >= 128 stores element e = tid - 128 of the previous step
    const int oe = tid - NE;
    const int ob = b0 + oe / UPC, ou = q * UPC + oe % UPC;
    const bool ovalid = tid >= NE && ob < B;
    const int olen = ovalid ? p.lens[ob] : 0;
    auto store_outputs = [&](int t) {                            // saved tensors of step t from outs[t & 1]
        if (!ovalid) return;
        const float* o = outs + (t & 1) * 6 * NE + oe;
        const size_t row = (size_t)t * LB + ob;
        p.y[row * p.ldy + ou] = o[0];
        if (p.gates && t < olen) {
            float* gp = p.gates + row * 4 * PH + ou;
            gp[0] = o[NE]; gp[(size_t)PH] = o[2 * NE]; gp[(size_t)2 * PH] = o[3 * NE]; gp[(size_t)3 * PH] = o[4 * NE];
            p.cell[row * PH + ou] = o[5 * NE];
        }
    };

    float c_state = 0.f, h_state = 0.f;
    // hand-off buffers of this group.  Tagged: two parity buffers of 8-byte granules.  BARE: three rotating buffers of bare
    // operand pairs behind ONE resource, the buffer of a step selected by a scalar byte offset.
    constexpr int LPG = BARE ? 1 : 2;                            // 16-byte loads per load group
    constexpr int DW_PER_GROUP = NCHUNK * 32 * RPGP / 2;         // BARE: dwords per buffer and group
    __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        rs[par] = __builtin_amdgcn_make_buffer_rsrc(p.hgran + ((size_t)par * NG + grp) * GRAN_PER_GROUP, 0,
                                                    GRAN_PER_GROUP * 8, 0x00020000);
    unsigned* const bare0 = reinterpret_cast<unsigned*>(p.hgran) + (size_t)grp * 3 * DW_PER_GROUP;
    const __amdgpu_buffer_rsrc_t rbare = __builtin_amdgcn_make_buffer_rsrc(bare0, 0, 3 * DW_PER_GROUP * 4, 0x00020000);
    int m3 = 0;                                                  // t % 3
    // load h of load group lg of this wave = 1 KiB at ((wave NLG + lg) LPG + h) KiB: lane offset in the VGPR, the rest scalar
    const int voff = lane * 16;
    const int soff_w = __builtin_amdgcn_readfirstlane(wave) * (NLG * LPG * 1024);
    const long t_start = wall_clock64();
    bool dead = false;
    // debug stamps (100 MHz wall clock): [step][wave][0..4] = loop top, sweep+MFMA done, reduce barrier passed, published,
    // passes of the poll loop
    const bool prof = p.prof != nullptr && grp == 0 && q == 0 && lane == 0;

    for (int t = 0; t < tg; ++t) {
        if ((t % SB) == 0) burst(t);
        long st0 = 0, st1 = 0, st2 = 0, st3 = 0, npass = 0;
        if (prof) st0 = wall_clock64();
        f32x4 acc[TPC];
#pragma unroll
        for (int j = 0; j < TPC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (WAGPR) {
            // zeroed HERE, in front of the gather: the compiler otherwise sinks each zeroing move to right in front of the inline-asm MFMA
            // that first reads the register -- a VALU write -> MFMA SrcC hazard it does not pad for an asm statement
#pragma unroll
            for (int j = 0; j < TPC; ++j) asm volatile("" : "+v"(acc[j]));
        }
        if (t > 0) {
            // ---- sweep: this wave's 8 chunks of h_{t-1} (epoch t) straight into A fragments.  Every pass re-reads ALL chunks
            // that have not shown the epoch yet (one L2 round trip for the lot).  The poll loop holds loads and tag compares
            // only; the 8 x TPC MFMAs follow as ONE straight-line block (MFMAs inside the data-dependent control flow made
            // hipcc shuffle accumulators and weight fragments through v_accvgpr_mov on every chunk).
            const unsigned epoch = (unsigned)t;
            const int par = (t - 1) & 1;
            const int boff = BARE ? (m3 == 0 ? 2 : m3 - 1) * (DW_PER_GROUP * 4) : 0;     // BARE: buffer (t - 1) % 3
            u32x4 ld[NLG][LPG];
            auto issue = [&](int g) {
#pragma unroll
                for (int h = 0; h < LPG; ++h)                                              // LAUX: 16 = sc1, 2 = nt
                    ld[g][h] = __builtin_amdgcn_raw_buffer_load_b128(BARE ? rbare : rs[par], voff, soff_w + (g * LPG + h) * 1024 + boff, LAUX);
            };
#pragma unroll
            for (int g = 0; g < NLG; ++g) issue(g);
            unsigned ready = 0;                    // wave-uniform bit per load group: every lane holds the data of this step
            for (unsigned spins = 0;; ++spins) {
#pragma unroll
                for (int g = 0; g < NLG; ++g) {
                    if (!((ready >> g) & 1u)) {
                        if (__all(fresh<BARE>(ld[g], epoch))) ready |= 1u << g;
                    }
                }
                npass = spins + 1;
                if (ready == (1u << NLG) - 1u) break;
                if ((spins & 15) == 15) {
                    if (wall_clock64() - t_start > p.timeout_ticks ||
                        __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        dead = true;
                        break;
                    }
                }
                asm volatile("" ::: "memory");     // the builtin loads are not atomics: keep the re-reads inside the loop
#pragma unroll
                for (int g = 0; g < NLG; ++g) {
                    if (!((ready >> g) & 1u)) issue(g);
                }
            }
            if (dead) break;
            if constexpr (WAGPR) {
                // chunk i of this wave = load group i / CPL, member i % CPL.  The four moves that assemble a chunk's A operand (payload
                // dwords of two loads, shifted into the MFMA row positions by DPP) are issued one per MFMA gap of the PREVIOUS chunk,
                // into the other of two operand buffers: in a block of their own they cost the one-wave-per-SIMD kernel their whole
                // issue time (4 moves + wait states per chunk, ~0.1 us per step)
                auto operand_word = [&](int i, int c) -> unsigned {
                    const u32x4 pl = payload<BARE>(ld[i / CPL]);
                    switch (i % CPL) {
                        case 0: {       // (an explicit move IN this slot: a plain copy is materialised by the compiler right in front of the
                            unsigned r; //  consuming asm MFMA -- VALU write -> MFMA source read without the two wait states: wrong operands)
                            asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(pl[c]));
                            return r;
                        }
                        case 1: return row_shl<RPGP & 15>(pl[c]);
                        case 2: return row_shl<(2 * RPGP) & 15>(pl[c]);
                        default: return row_shl<(3 * RPGP) & 15>(pl[c]);
                    }
                };
                u32x4 au[3];                                                      // three buffers: the one being assembled was last read a whole
#pragma unroll                                                                    // chunk (8 MFMAs) ago
                for (int c = 0; c < 4; ++c) au[0][c] = operand_word(0, c);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < TPC; ++j) {
                        if (i == 0 && j == 0) mfma16_bagpr_nop(acc[0], au[0], __builtin_bit_cast(u32x4, w[0][0]));   // (two wait states behind the moves above)
                        else mfma16_bagpr(acc[j], au[i % 3], __builtin_bit_cast(u32x4, w[j][i]));
                        if (i + 1 < 8 && j < 4) {
                            __builtin_amdgcn_sched_barrier(0);
                            au[(i + 1) % 3][j] = operand_word(i + 1, j);
                            __builtin_amdgcn_sched_barrier(0);
                        }

                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, member<RPGP>(payload<BARE>(ld[i / CPL]), i % CPL));
#pragma unroll
                    for (int j = 0; j < TPC; ++j) acc[j] = mfma16(a, w[j][i], acc[j]);
                }
            }
            // (asm MFMAs: the compiler does not know their result latency -- 12 wait states before anything reads the last one's)
            if constexpr (WAGPR) asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
        }
        // D[m = batch row (lane>>4)*4 + r][n = li]: rows >= RPGP are padding
        const int rb = t & 1;
        if (prof) st1 = wall_clock64();
        if (kg * 4 < RPGP) {
#pragma unroll
            for (int j = 0; j < TPC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[rb][wave][j][kg * 4 + r][li] = acc[j][r];
        }
        __syncthreads();
        if (prof) st2 = wall_clock64();
        if (t > 0) store_outputs(t - 1);
        if (erole) {
            const bool active = t < len;
            const int j = el >> 2, ul = el & 3;
            if constexpr (BARE) {
                // reset this thread's slot of buffer (t + 1) % 3 (it holds step t - 2, which everybody has consumed) first: the
                // acknowledgement returns under the cell update below, and the publish waits for it
                if ((el & 1) == 0) {
                    unsigned* dst = bare0 + (m3 == 2 ? 0 : m3 + 1) * DW_PER_GROUP + bare_index<RPGP, 8>(ebl, eu);
                    if constexpr (LOCAL) __hip_atomic_store((gu32*)dst, SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store((gu32*)dst, SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const float* gxr = gxs + (t % SB) * 4 * NE + tid;
            float pre[4];
            {
                // the four gates of (unit ul, row ebl) are adjacent (make_wfrag_fwd_ug): one 16-byte read per wave partial; the sum
                // keeps its order (wave 0 + 1 + 2 + 3, then gx)
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(&red[rb][0][j][ebl][ul * 4]), p1 = *reinterpret_cast<const f32x4*>(&red[rb][1][j][ebl][ul * 4]);
                const f32x4 p2 = *reinterpret_cast<const f32x4*>(&red[rb][2][j][ebl][ul * 4]), p3 = *reinterpret_cast<const f32x4*>(&red[rb][3][j][ebl][ul * 4]);
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = p0[g] + p1[g] + p2[g] + p3[g] + gxr[g * NE];
            }
            float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f;
            if (active) {
                float c_new, h_new;
                lstm_cell<true>(pre, c_state, ig, fg, gg, og, c_new, h_new);
                c_state = c_new; h_state = h_new;
            }
            // ---- publish h_t first: one 8-byte {epoch, bf16 pair} granule per even unit (frozen rows re-publish their state)
            const float h_nb = __uint_as_float(row_shl<1>(__float_as_uint(h_state)));     // lane + 1 of the row: the odd unit of the pair
            if constexpr (BARE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the reset above is in the L2 before the publish leaves
                if ((el & 1) == 0) {
                    unsigned* dst = bare0 + m3 * DW_PER_GROUP + bare_index<RPGP, 8>(ebl, eu);
                    const unsigned val = pack_op16x2(h_state, h_nb);
                    if constexpr (LOCAL) __hip_atomic_store((gu32*)dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store((gu32*)dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if ((el & 1) == 0) {
                const unsigned long long gran = ((unsigned long long)(unsigned)(t + 1) << 32) | pack_op16x2(h_state, h_nb);
                unsigned long long* dst = p.hgran + ((size_t)(t & 1) * NG + grp) * GRAN_PER_GROUP + gran_index<RPGP, 8>(ebl, eu);
                // LOCAL: workgroup-scope relaxed store = ONE aligned 8-byte global_store (sc0) whose line stays in this XCD's L2;
                // otherwise agent scope = sc1, write-through to the memory side
                if constexpr (LOCAL) __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (prof) st3 = wall_clock64();
            float* o = outs + (t & 1) * 6 * NE + tid;            // waves 2-3 store it during the next step
            o[0] = active ? h_state : 0.f;
            o[NE] = ig; o[2 * NE] = fg; o[3 * NE] = gg; o[4 * NE] = og; o[5 * NE] = c_state;
        }
        if (prof && t < 1024) {
            long* o = p.prof + ((size_t)t * 4 + wave) * 5;
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = npass;
        }
        m3 = m3 == 2 ? 0 : m3 + 1;
    }
    if (dead) {
        if (lane == 0) atomicExch(p.status, 1);
        return;
    }
    __syncthreads();
    if (tg > 0) store_outputs(tg - 1);
    // pad rows beyond the group's longest sequence: y = 0 (pad_packed_sequence semantics)
    if (ev)
        for (int t = tg; t < T; ++t) p.y[((size_t)t * LB + eb) * p.ldy + eu] = 0.f;
}


// ---------------------------------------------------------------------------------------------------------------
// Backward recurrence, same organisation: dh_rec[b][j] = sum_r dgates_{s+1}[b][r] W_hh[r][j]  (K = 4H, N = H).
// CU q of a group owns UPC hidden units j (UPC/16 column tiles of W_hh^T, fragments resident in registers: wave w holds
// k-chunks w, w+4, .. of K = 4H = 128 chunks) and runs the cell backward of those units for the group's rows, publishing
// dgates_s (4 gates x UPC units x RPGP rows) as granules over k = gate*H + j.  The group's dgates vector is 4x the forward
// state (RPGP x 4H bf16), so a wave sweeps its 32 chunks in 4 batches of 8 with the next batch's loads in flight.
// Accumulation mimics lstm_bwd_step_bf16's 16-wave split (partial a of wave w = chunks w+4a, w+4a+16, ..; the 16 partials
// are summed in wave order), so the result is bit-identical to the launch-per-step kernel.  Saved gates / cell / dy come
// in through the same LDS ring as the forward kernel (dgx goes out from waves 2-3).
struct PersistBwdP {
    const float* dy; long ldy; const int* lens;
    const float* gates; const float* cell; float* dgx;
    const unsigned short* wTfrag;        // [H/16][4H/32][64][8] bf16 (make_wfrag_bwd layout)
    unsigned long long* dgran;           // [2 parity][NG][4H/32][4 kg][RPGP][4] granules
    int* status; unsigned* census;
    int T, B;
    long timeout_ticks;
    long* prof;
    // optional (dimg != nullptr): the 16-bit operand image of dgates in pack-by-length row order -- what ft_bf16_image_rows would
    // make of dgx afterwards (batch-major compact rows, utterance b = its len_b frames + one zero separator row, zero rows up to
    // ceil256(R + 32)) -- written by the output waves beside the fp32 rows, and the column sums of dgates (the bias gradient)
    // added to dbias[4H]: the weight-gradient / input-gradient GEMMs then start without a conversion pass over dgx.
    unsigned short* dimg; long dimg_ld; int dimg_rows; float* dbias;
    int LB;                              // batch rows per time step in memory (see PersistP; the image output needs LB == B)
};

// OUT: 0 = fp32 dgx rows only, 1 = dgx and the compact 16-bit image, 2 = the image only (compile-time: the plain path carries none
// of the image code, the image-only path none of the fp32 stores)
template <int NG, bool LOCAL, int LAUX, bool BARE, int OUT = 0>
__global__ __launch_bounds__(256, 1) void lstm_persist_bwd_k(PersistBwdP p) {
    constexpr bool WF32 = OUT != 2, WIMG = OUT != 0;
    static_assert(!LOCAL || NG == 8, "the L2-local transport needs group == XCD");
    constexpr int CPG = NCU / NG;                  // workgroups per group
    constexpr int UPC = PH / CPG;                  // hidden units per CU: 32 (NG 8), 16 (NG 4)
    constexpr int TL = UPC / 16;                   // column tiles per CU
    constexpr int RPGP = 32 / NG;
    constexpr int NE = RPGP * UPC;
    constexpr int KCH = 4 * PH / 32;               // 128 k-chunks
    constexpr int CPW = KCH / 4;                   // chunks per wave: 32
    constexpr int GRAN_PER_GROUP = KCH * 4 * RPGP * 4;
    constexpr int CPL = 16 / RPGP, NLG = CPW / CPL; // chunks per load, load groups per wave (x 2 halves): 8 (NG 8), 16 (NG 4)
    constexpr int NBT = NLG > 8 ? 2 : 1, LPB = NLG / NBT;   // sweep batches per step (<= 16 loads in flight each), load groups per batch
    static_assert(TL >= 1, "NG = 2 would leave half a column tile per CU");
    static_assert(NE == 128, "one epilogue element per thread of waves 0-1");
    // LDS: 16-partial reduce (double buffered) | RING steps in [slot][gates x4, dy][e] | RING cells [slot][e] | 2 steps out [parity][4][e]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*red)[16][TL][RPGP][17] = reinterpret_cast<float (*)[16][TL][RPGP][17]>(smem);
    float* ins = smem + 2 * 16 * TL * RPGP * 17;
    float* cells = ins + RING * 5 * NE;
    float* outs = cells + RING * NE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int grp, q;
    if constexpr (LOCAL) {
        if (!join_group_local<CPG>(p.census, p.status, tid, grp, q)) return;
    } else {
        grp = blockIdx.x % NG; q = blockIdx.x / NG;
    }
    const int B = p.B, T = p.T, LB = p.LB;
    const int b0 = grp * RPGP;

    bf16x8 w[TL][CPW];
    {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.wTfrag);
#pragma unroll
        for (int j = 0; j < TL; ++j)
#pragma unroll
            for (int i = 0; i < CPW; ++i) w[j][i] = wf[((size_t)(q * TL + j) * KCH + (wave + 4 * i)) * 64 + lane];
    }

    const bool erole = tid < NE;
    const int el = tid % UPC, ebl = tid / UPC;
    const int eb = b0 + ebl, eu = q * UPC + el;
    const bool ev = erole && eb < B;
    const int len = ev ? p.lens[eb] : 0;
    int tg = 0;
#pragma unroll
    for (int r = 0; r < RPGP; ++r) {
        const int bb = b0 + r;
        if (bb < B) { const int l = p.lens[bb]; tg = l > tg ? l : tg; }
    }
    tg = tg < T ? tg : T;

    // step counter n = 0 .. tg-1 walks time s = tg-1-n downwards
    // output role of waves 2-3 (see the forward kernel): thread tid >= 128 stores the dgx row of the previous step
    const int oe = tid - NE;
    const int ob = b0 + oe / UPC, ou = q * UPC + oe % UPC;
    const bool ovalid = tid >= NE && ob < B;
    int olen = 0, ooff = 0;                                      // image: this row's length and its first compact row
    if (WIMG && ovalid) {
        olen = p.lens[ob];
        for (int bb = 0; bb < ob; ++bb) ooff += p.lens[bb] + 1;
    }
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto store_outputs = [&](int n) {                            // dgx row of step counter n from outs[n & 1]
        if (!ovalid) return;
        const float* o = outs + (n & 1) * 4 * NE + oe;
        const int so = tg - 1 - n;
        const float v0 = o[0], v1 = o[NE], v2 = o[2 * NE], v3 = o[3 * NE];
        if constexpr (WF32) {
            float* dg = p.dgx + ((size_t)so * LB + ob) * 4 * PH + ou;
            dg[0] = v0; dg[(size_t)PH] = v1; dg[(size_t)2 * PH] = v2; dg[(size_t)3 * PH] = v3;
        }
        if (WIMG && so < olen) {
            unsigned short* ip = p.dimg + (size_t)(ooff + so) * p.dimg_ld + ou;
            ip[0] = (unsigned short)pack_op16x2(v0, 0.f); ip[PH] = (unsigned short)pack_op16x2(v1, 0.f);
            ip[2 * PH] = (unsigned short)pack_op16x2(v2, 0.f); ip[3 * PH] = (unsigned short)pack_op16x2(v3, 0.f);
            bsum[0] += v0; bsum[1] += v1; bsum[2] += v2; bsum[3] += v3;
        }
    };
    // ring slot n % RING: saved gates x4 + dy of step n; cell slot n % RING = cell[s(n) - 1] (c_prev of step n = c_t of step n+1),
    // cell slot RING-1 starts out as cell[tg - 1] (c_t of step 0)
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int eh = (wave & 1) * 64 + lane;
    const int hb = b0 + eh / UPC, hu = q * UPC + eh % UPC;
    const bool hvalid = hb < B;
    const unsigned ins0 = (unsigned)(size_t)(lds_void*)ins + (unsigned)(wu & 1) * 256u;
    const unsigned cells0 = (unsigned)(size_t)(lds_void*)cells + (unsigned)(wu & 1) * 256u;
    auto prefetch = [&](int m) {                                 // waves 2-3 (from the epilogue waves instead: 3.13 vs 3.07 us)
        if (wu >= 2 && m < tg && hvalid) {
            const int sm = tg - 1 - m;
            const size_t row = (size_t)sm * LB + hb;
            const unsigned dst = ins0 + (unsigned)((m % RING) * 5 * NE * 4);
#pragma unroll
            for (int f = 0; f < 4; ++f) dma_dword(p.gates + row * 4 * PH + (size_t)f * PH + hu, dst + (unsigned)(f * NE * 4));
            dma_dword(p.dy + row * p.ldy + hu, dst + (unsigned)(4 * NE * 4));
            if (sm > 0) dma_dword(p.cell + (row - LB) * PH + hu, cells0 + (unsigned)((m % RING) * NE * 4));
        }
    };
    if (wu >= 2 && tg > 0 && hvalid) dma_dword(p.cell + ((size_t)(tg - 1) * LB + hb) * PH + hu, cells0 + (unsigned)((RING - 1) * NE * 4));
#pragma unroll
    for (int m = 0; m < DIST; ++m) prefetch(m);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float dc_carry = 0.f;
    constexpr int LPG = BARE ? 1 : 2;                            // 16-byte loads per load group (see the forward kernel)
    constexpr int DW_PER_GROUP = KCH * 32 * RPGP / 2;            // BARE: dwords per buffer and group
    __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        rs[par] = __builtin_amdgcn_make_buffer_rsrc(p.dgran + ((size_t)par * NG + grp) * GRAN_PER_GROUP, 0,
                                                    GRAN_PER_GROUP * 8, 0x00020000);
    unsigned* const bare0 = reinterpret_cast<unsigned*>(p.dgran) + (size_t)grp * 3 * DW_PER_GROUP;
    const __amdgpu_buffer_rsrc_t rbare = __builtin_amdgcn_make_buffer_rsrc(bare0, 0, 3 * DW_PER_GROUP * 4, 0x00020000);
    int m3 = 0;                                                  // n % 3
    const int voff = lane * 16;
    const int soff_w = __builtin_amdgcn_readfirstlane(wave) * (NLG * LPG * 1024);
    const long t_start = wall_clock64();
    bool dead = false;

    // debug stamps: [step][wave][0..4] = loop top, polls + MFMAs done, partials written + barrier passed, published, poll passes
    const bool prof = p.prof != nullptr && grp == 0 && q == 0 && lane == 0;
    for (int n = 0; n < tg; ++n) {                 // n-th step of the sweep: time index s = tg-1-n
        long st0 = 0, st1 = 0, st2 = 0, st3 = 0, npass = 0;
        if (prof) st0 = wall_clock64();
        const int s = tg - 1 - n;
        f32x4 acc[TL][4];
#pragma unroll
        for (int j = 0; j < TL; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[j][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (n > 0) {
            const unsigned epoch = (unsigned)n;
            const int par = (n - 1) & 1;
            const int boff = BARE ? (m3 == 0 ? 2 : m3 - 1) * (DW_PER_GROUP * 4) : 0;     // BARE: buffer (n - 1) % 3
            u32x4 ld[2][LPB][LPG];
            auto issue1 = [&](int bt, int g, u32x4 (&l_)[LPG]) {
#pragma unroll
                for (int h = 0; h < LPG; ++h)
                    l_[h] = __builtin_amdgcn_raw_buffer_load_b128(BARE ? rbare : rs[par], voff, soff_w + ((bt * LPB + g) * LPG + h) * 1024 + boff, LAUX);
            };
            auto issue = [&](int bt, u32x4 (&l_)[LPB][LPG]) {
#pragma unroll
                for (int g = 0; g < LPB; ++g) issue1(bt, g, l_[g]);
            };
            issue(0, ld[0]);
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) {
                if (bt < NBT - 1) issue(bt + 1, ld[(bt + 1) & 1]);
                u32x4 (&L_)[LPB][LPG] = ld[bt & 1];
                unsigned ready = 0;
                for (unsigned spins = 0;; ++spins) {             // loads and tag compares only (see the forward kernel)
#pragma unroll
                    for (int g = 0; g < LPB; ++g) {
                        if (!((ready >> g) & 1u)) {
                            if (__all(fresh<BARE>(L_[g], epoch))) ready |= 1u << g;
                        }
                    }
                    if (ready == (1u << LPB) - 1u) break;
                    if (prof) ++npass;
                    if ((spins & 15) == 15) {
                        if (wall_clock64() - t_start > p.timeout_ticks ||
                            __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            dead = true;
                            break;
                        }
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int g = 0; g < LPB; ++g) {
                        if (!((ready >> g) & 1u)) issue1(bt, g, L_[g]);
                    }
                }
                if (dead) break;
#pragma unroll
                for (int i = 0; i < LPB * CPL; ++i) {                // chunk ci of this wave = load group ci / CPL, member ci % CPL
                    const int ci = bt * LPB * CPL + i;
                    const bf16x8 a = __builtin_bit_cast(bf16x8, member<RPGP>(payload<BARE>(L_[i / CPL]), i % CPL));
#pragma unroll
                    for (int j = 0; j < TL; ++j)
                        acc[j][ci & 3] = mfma16(a, w[j][ci], acc[j][ci & 3]);
                }
            }
            if (dead) break;
        }
        const int rb = n & 1;
        if (prof) st1 = wall_clock64();
        if (kg * 4 < RPGP) {
#pragma unroll
            for (int j = 0; j < TL; ++j)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[rb][wave + 4 * a][j][kg * 4 + r][li] = acc[j][a][r];
        }
        __syncthreads();
        if (prof) st2 = wall_clock64();
        if (n > 0) store_outputs(n - 1);
        prefetch(n + DIST);
        // the output waves' next poll could not be consumed before these stores / DMAs have landed anyway (vmcnt retires in
        // order); issued now it would read the granules BEFORE the group has published and cost a second round trip
        if (wu >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (erole) {
            const int i = n % RING;
            const bool active = s < len;
            float da[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BARE) {
                // reset this thread's four slots of buffer (n + 1) % 3 (step n - 2: consumed by everybody); acknowledged under the
                // cell backward below (waves 0-1 have nothing else outstanding: the ring DMAs and dgx stores belong to waves 2-3)
                if ((el & 1) == 0) {
                    unsigned* base = bare0 + (m3 == 2 ? 0 : m3 + 1) * DW_PER_GROUP;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned* dst = base + bare_index<RPGP, CPW>(ebl, g * PH + eu);
                        if constexpr (LOCAL) __hip_atomic_store((gu32*)dst, SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else __hip_atomic_store((gu32*)dst, SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (active) {
                const int j = el >> 4, nn = el & 15;
                const float* in = ins + i * 5 * NE + tid;
                float dh = in[4 * NE];
#pragma unroll
                for (int w16 = 0; w16 < 16; ++w16) dh += red[rb][w16][j][ebl][nn];
                const float c_t = cells[((n + RING - 1) % RING) * NE + tid], c_prev = s > 0 ? cells[i * NE + tid] : 0.f;
                float carry;
                lstm_cell_bwd<true>(dh, dc_carry, in[0], in[NE], in[2 * NE], in[3 * NE], c_t, c_prev, da, carry);
                dc_carry = carry;
            }
            // ---- publish dgates_s: one granule per gate per even unit (k = gate*H + unit)
            if constexpr (BARE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the resets above have reached the L2
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float nb = __uint_as_float(row_shl<1>(__float_as_uint(da[g])));
                if ((el & 1) == 0) {
                    if constexpr (BARE) {
                        unsigned* dst = bare0 + m3 * DW_PER_GROUP + bare_index<RPGP, CPW>(ebl, g * PH + eu);
                        const unsigned val = pack_op16x2(da[g], nb);
                        if constexpr (LOCAL) __hip_atomic_store((gu32*)dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else __hip_atomic_store((gu32*)dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        const unsigned long long gran = ((unsigned long long)(unsigned)(n + 1) << 32) | pack_op16x2(da[g], nb);
                        unsigned long long* dst = p.dgran + ((size_t)(n & 1) * NG + grp) * GRAN_PER_GROUP + gran_index<RPGP, CPW>(ebl, g * PH + eu);
                        if constexpr (LOCAL) __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else __hip_atomic_store((gu64*)dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (prof) st3 = wall_clock64();
            float* o = outs + (n & 1) * 4 * NE + tid;
#pragma unroll
            for (int g = 0; g < 4; ++g) o[g * NE] = da[g];
        }
        if (prof && n < 1024) {
            long* o = p.prof + ((size_t)n * 4 + wave) * 5;
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = npass;
        }
        m3 = m3 == 2 ? 0 : m3 + 1;
    }
    if (dead) {
        if (lane == 0) atomicExch(p.status, 1);
        return;
    }
    __syncthreads();
    if (tg > 0) store_outputs(tg - 1);
    if (WF32 && ev)                                                     // pad rows beyond the group's longest sequence
        for (int t = tg; t < T; ++t) {
            float* dg = p.dgx + ((size_t)t * LB + eb) * 4 * PH + eu;
#pragma unroll
            for (int g = 0; g < 4; ++g) dg[(size_t)g * PH] = 0.f;
        }
    if constexpr (WIMG) {
        if (ovalid) {                                                   // bias gradient; the utterance's zero separator row
            unsigned short* ip = p.dimg + (size_t)(ooff + olen) * p.dimg_ld + ou;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                atomicAdd(p.dbias + g * PH + ou, bsum[g]);
                ip[g * PH] = 0;
            }
        }
        // zero rows behind the last utterance up to ceil256(R + 32) (gemm_bf16.hip mapped_rows): workgroup w takes rows R + w, R + w + 256
        int R = 0;
        for (int bb = 0; bb < B; ++bb) R += p.lens[bb] + 1;
        int Rz = (R + 32 + 255) & ~255;
        Rz = Rz < p.dimg_rows ? Rz : p.dimg_rows;
        for (int r = R + grp * CPG + q; r < Rz; r += NCU) {
            uint4* row = reinterpret_cast<uint4*>(p.dimg + (size_t)r * p.dimg_ld);
            for (int c = tid; c < 4 * PH / 8; c += 256) row[c] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Backward recurrence, REDUCE-SCATTER form (round 4).  The all-gather form above hands the group's dgates vector (RPGP x 4H
// 16-bit operands = 32 KB per CU and step) to every CU, which then multiplies it with ITS 32 columns of W_hh^T: measured, ~0.6 us
// of its 2.8 us step are the 1 MB per XCD and step the 32 CUs pull out of their L2, 0.2 us the sentinel resets.  Here the product
// is split the other way:
//   CU q keeps the W_hh rows of ITS OWN 128 gate rows (c = gate*32 + unit; the forward kernel's slice, [128 x H] as B fragments:
//        wave w holds column tiles 16 w .. 16 w + 15 for the four k-chunks = gates) and multiplies its own, LOCAL dgates
//        (RPGP x 128, staged through 1 KB of LDS -- no all-gather at all) into a partial dh_rec [RPGP x H] in fp32;
//   the partials are REDUCE-SCATTERED through the XCD's L2: CU q publishes, for every consumer q', its [32 units x RPGP rows] slab
//        (512 B, one 16-byte store per lane and tile), and gathers the 32 slabs addressed to it (16 KB per CU and step: half the
//        bytes of the bare all-gather, a quarter of the tagged one), summing them in a fixed order: 4 loads per wave in registers,
//        then 8 per-wave-half sums through LDS.
// Hand-off: every fp32 dword carries its own tag in the mantissa LSB (tag = bit 1 of the step counter; two parity buffers, so a
// slot alternates tag per reuse; initial fill 0xFF = tag 1 != tag of steps 0 / 1): no sentinel resets, no epoch words, no extra
// bytes -- the payload loses one mantissa bit (relative 2^-23, four orders below the 16-bit operand rounding of dgates).
// The sum over the 32 partials is NOT the launch-per-step kernel's summation order: results agree with lstm_bwd_step_bf16 to fp32
// rounding (tests: tolerance, not bit-identity).  Roles: all four waves poll, multiply and publish; between the two barriers of
// a step waves 0-1 run the cell backward while waves 2-3 store the previous step's dgates rows / image and issue the ring DMAs.
#ifndef FT_RS_TAGS
#define FT_RS_TAGS 2
#endif
template <int OUT, bool PROF = false>
__global__ __launch_bounds__(256, 1) void lstm_persist_bwd_rs_k(PersistBwdP p) {
    // Tags per 16-byte slab piece (one lane's store = the four rows of one unit): 4 = every dword carries its own (nothing assumed about
    // how a 16-byte store becomes visible), 2 = the FIRST and the LAST dword (default: a store observed torn at any single split
    // point fails the check; measured 0.1 us per step cheaper than 4 -- a tag costs a VALU instruction per dword on the publish side,
    // issue slots the MFMA block does not hide), 1 = the first dword only (relies on an aligned 16-byte store never being observed
    // torn by an aligned 16-byte load: what cdna_hip_programming.md G16 reports for gfx950 without calling it a guarantee; 1.73 us).
    constexpr int NTAG = FT_RS_TAGS;
    auto tagged = [](int r) { return NTAG == 4 || r == 0 || (NTAG == 2 && r == 3); };
    constexpr bool WF32 = OUT != 2, WIMG = OUT != 0;
    constexpr int NG = 8, CPG = NCU / NG, UPC = PH / CPG, RPGP = 32 / NG, NE = RPGP * UPC;      // 32 CUs, 32 units, 4 rows, 128 elements
    constexpr int NT = PH / 16 / 4;                // column tiles per wave: 16
    static_assert(UPC == 32 && RPGP == 4 && NE == 128, "built for group == XCD: 32 CUs x 32 units, 4 batch rows");
    // LDS: gather sums [8 = wave x half][32 units][4 rows] | dgates operands [4 gates][16 A-tile rows][32 units] 16-bit (rows >= RPGP
    // zero for good: every lane reads its fragment without a branch) | RING steps in [slot][gates x4, dy, previous cell][e] |
    // 2 steps out [parity][4][e]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gsum = smem;                                             // 8 * 128 floats
    unsigned* daop = reinterpret_cast<unsigned*>(smem + 8 * NE);    // 4 * 16 * 16 dwords (pairs of 16-bit operands)
    float* ins = smem + 8 * NE + 1024;                              // ring: [slot][gates x4, dy, cell of the step before][e]
    float* outs = ins + RING * 6 * NE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    int grp, q;
    if (!join_group_local<CPG>(p.census, p.status, tid, grp, q)) return;
    const int B = p.B, T = p.T, LB = p.LB;
    const int b0 = grp * RPGP;
    for (int i2 = tid; i2 < 1024; i2 += 256) daop[i2] = 0u;          // (made visible by the barrier behind the first ring fill)

    // ---- resident weights: tile 16 wave + j, chunk (= gate) g
    u32x4 w[NT][4];
    {
        const u32x4* wf = reinterpret_cast<const u32x4*>(p.wTfrag);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) w[j][g] = wf[((size_t)((q * (PH / 16) + wave * NT + j) * 4 + g)) * 64 + lane];
    }

    const bool erole = tid < NE;
    const int el = tid % UPC, ebl = tid / UPC;
    const int eb = b0 + ebl, eu = q * UPC + el;
    const bool ev = erole && eb < B;
    const int len = ev ? p.lens[eb] : 0;
    int tg = 0;
#pragma unroll
    for (int r = 0; r < RPGP; ++r) {
        const int bb = b0 + r;
        if (bb < B) { const int l = p.lens[bb]; tg = l > tg ? l : tg; }
    }
    tg = tg < T ? tg : T;

    // output role of waves 2-3 (as in lstm_persist_bwd_k)
    const int oe = tid - NE;
    const int ob = b0 + oe / UPC, ou = q * UPC + oe % UPC;
    const bool ovalid = tid >= NE && ob < B;
    int olen = 0, ooff = 0;
    if (WIMG && ovalid) {
        olen = p.lens[ob];
        for (int bb = 0; bb < ob; ++bb) ooff += p.lens[bb] + 1;
    }
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto store_outputs = [&](int n) {
        if (!ovalid) return;
        const float* o = outs + (n & 1) * 4 * NE + oe;
        const int so = tg - 1 - n;
        const float v0 = o[0], v1 = o[NE], v2 = o[2 * NE], v3 = o[3 * NE];
        if constexpr (WF32) {
            float* dg = p.dgx + ((size_t)so * LB + ob) * 4 * PH + ou;
            dg[0] = v0; dg[(size_t)PH] = v1; dg[(size_t)2 * PH] = v2; dg[(size_t)3 * PH] = v3;
        }
        if (WIMG && so < olen) {
            unsigned short* ip = p.dimg + (size_t)(ooff + so) * p.dimg_ld + ou;
            ip[0] = (unsigned short)pack_op16x2(v0, 0.f); ip[PH] = (unsigned short)pack_op16x2(v1, 0.f);
            ip[2 * PH] = (unsigned short)pack_op16x2(v2, 0.f); ip[3 * PH] = (unsigned short)pack_op16x2(v3, 0.f);
            bsum[0] += v0; bsum[1] += v1; bsum[2] += v2; bsum[3] += v3;
        }
    };
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    // Ring fill: a slot is 6 rows x 128 floats = three 1 KiB LDS-DMA pieces (global_load_lds_dwordx4: 16 bytes per lane, eight lanes per
    // 128-byte row piece of one batch row) instead of twelve 256-byte ones -- a piece costs 60-185 cycles of issue time whatever its
    // size, and the two output waves that issue them gate the step's second barrier.  Wave 2 moves the gate rows (pieces 0, 1), wave 3
    // the dy row and the previous step's cell row (piece 2).  Piece k = rows 2k, 2k + 1; lane -> row 2k + (lane >> 5), batch row
    // (lane & 31) >> 3, units 4 (lane & 7) .. + 3.  (16-byte alignment of dy rows is checked by the launcher: else the dword path.)
    const int pr = lane >> 5, pb = (lane & 31) >> 3, pu = 4 * (lane & 7);
    const bool pvalid = b0 + pb < B;
    const unsigned ring0 = (unsigned)(size_t)(lds_void*)ins;
    auto dma16 = [&](const float* src, unsigned lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr)) : "memory");
    };
    auto prefetch = [&](int m) {
        if (wu < 2 || m >= tg) return;
        const int sm = tg - 1 - m;
        const size_t row = (size_t)sm * LB + b0 + pb;
        const unsigned dst = ring0 + (unsigned)((m % RING) * 6 * NE * 4);
        if (wu == 2) {
            if (pvalid) {
                dma16(p.gates + row * 4 * PH + (size_t)pr * PH + q * UPC + pu, dst);                       // gates i, f
                dma16(p.gates + row * 4 * PH + (size_t)(2 + pr) * PH + q * UPC + pu, dst + 2 * NE * 4);      // gates g, o
            }
        } else {
            // piece 2: row 4 = dy of step sm, row 5 = cell of step sm - 1 (absent for sm == 0: those lanes stay out)
            const float* src = pr == 0 ? p.dy + row * p.ldy + q * UPC + pu : p.cell + (row - LB) * PH + q * UPC + pu;
            if (pvalid && (pr == 0 || sm > 0)) dma16(src, dst + 4 * NE * 4);
        }
    };
    // the cell of the LAST step (c_t of n = 0) goes to row 5 of slot RING - 1: upper half of a piece-2 DMA by wave 3
    if (wu == 3 && tg > 0 && pvalid && pr == 1)
        dma16(p.cell + ((size_t)(tg - 1) * LB + b0 + pb) * PH + q * UPC + pu, ring0 + (unsigned)(((RING - 1) * 6 + 4) * NE * 4));
#pragma unroll
    for (int m = 0; m < DIST; ++m) prefetch(m);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and the compiler is TOLD so (the builtin is visible to its wait-count pass, the asm is not): otherwise the first use of every
    // register loaded above -- the weight fragments at the first MFMA, the lengths in the output path -- sits inside the loop behind a
    // vmcnt(0) that, executed every step, waits for whatever is in flight then: the ring DMAs and the output stores of waves 2-3
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+a"(w[j][g]));   // every fragment lives in accumulation registers from here on
    asm volatile("" :: "v"(olen), "v"(ooff), "v"(len));
    __syncthreads();

    float dc_carry = 0.f;
    // partial buffers: [2 parity][NG][consumer 32][producer 32][32 units][4 rows] fp32.  This CU reads its consumer block (16 KB,
    // wave w the producers 8 w .. 8 w + 7 = 4 loads of 1 KB) and writes slab [consumer][q] of every consumer block.
    constexpr size_t PB_GROUP = (size_t)CPG * CPG * UPC * RPGP;      // floats per group and parity (131 072 = 512 KB)
    float* const pbase = reinterpret_cast<float*>(p.dgran);
    __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        rs[par] = __builtin_amdgcn_make_buffer_rsrc(pbase + ((size_t)par * NG + grp) * PB_GROUP + (size_t)q * CPG * UPC * RPGP, 0,
                                                    CPG * UPC * RPGP * 4, 0x00020000);
    const int voff = lane * 16;
    const int soff_w = __builtin_amdgcn_readfirstlane(wave) * 4096;
    // write side: the whole parity buffer of the group; lane offset = this producer's slab + its unit, scalar offset = consumer block
    __amdgpu_buffer_rsrc_t wrs[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        wrs[par] = __builtin_amdgcn_make_buffer_rsrc(pbase + ((size_t)par * NG + grp) * PB_GROUP, 0, (int)(PB_GROUP * 4), 0x00020000);
    const int wvoff = kg == 0 ? (q * UPC * RPGP + li * RPGP) * 4 : (int)0x7ffffff0;      // (beyond num_records: dropped)
    const int wsoff_w = __builtin_amdgcn_readfirstlane(wave) * 8 * (CPG * UPC * RPGP * 4);
    const long t_start = wall_clock64();
    bool dead = false;
    // phase stamps only in the PROF instantiation (ft_lstm_persist_debug_prof): their per-step stores make the compiler wait for
    // ALL outstanding memory operations -- ring DMAs included -- where it reuses the stamp registers
    const bool prof = PROF && p.prof != nullptr && grp == 0 && q == 0 && lane == 0;

    for (int n = 0; n < tg; ++n) {
        long st0 = 0, st1 = 0, st2 = 0, st3 = 0, npass = 0;
        if (prof) st0 = wall_clock64();
        const int s = tg - 1 - n;
        // everything of the cell backward that does not need dh_rec -- the ring reads, tanh(c_t), the five gate-derivative factors --
        // is evaluated by the epilogue waves while the gather loads are in flight:
        //   dc = dh fA + dc_carry ; carry' = dc f ; da_i = dc fI ; da_f = dc fF ; da_g = dc fG ; da_o = dh fO
        //   fA = o (1 - tanh^2 c_t), fO = tanh(c_t) o (1 - o), fI = g i (1 - i), fF = c_prev f (1 - f), fG = i (1 - g^2)
        // (lstm_cell_bwd's products re-associated: equal to fp32 rounding, like the partial sums themselves)
        float fA = 0.f, fO = 0.f, fI = 0.f, fF = 0.f, fG = 0.f, fgate = 0.f, dy_s = 0.f;
        const bool active = erole && s < len;
        auto precompute = [&]() {
            if (active) {
                const int i = n % RING;
                const float* in = ins + i * 6 * NE + tid;
                const float ig = in[0], fg = in[NE], gg = in[2 * NE], og = in[3 * NE];
                dy_s = in[4 * NE];
                const float c_t = ins[(((n + RING - 1) % RING) * 6 + 5) * NE + tid], c_prev = s > 0 ? in[5 * NE] : 0.f;
                const float tc = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * 1.4426950408889634f * c_t) + 1.f);
                fA = og * __fmaf_rn(-tc, tc, 1.f);
                fO = tc * og * (1.f - og);
                fI = gg * ig * (1.f - ig);
                fF = c_prev * fg * (1.f - fg);
                fG = ig * __fmaf_rn(-gg, gg, 1.f);
                fgate = fg;
            }
        };
        if (n > 0) {
            // ---- gather: the partials of step n - 1 addressed to this CU (buffer (n-1) & 1, tag = bit 1 of n - 1)
            const unsigned tag = (unsigned)((n - 1) >> 1) & 1u;
            const int par = (n - 1) & 1;
            u32x4 ld[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) ld[g] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff, soff_w + g * 1024, 2);
            precompute();
            unsigned ready = 0;
            for (unsigned spins = 0;; ++spins) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (!((ready >> g) & 1u)) {
                        const bool ok = NTAG == 1 ? (((ld[g][0] ^ tag) & 1u) == 0u)
                                        : NTAG == 2 ? ((((ld[g][0] ^ tag) | (ld[g][3] ^ tag)) & 1u) == 0u)
                                                    : ((((ld[g][0] ^ tag) | (ld[g][1] ^ tag) | (ld[g][2] ^ tag) | (ld[g][3] ^ tag)) & 1u) == 0u);
                        if (__all(ok)) ready |= 1u << g;
                    }
                }
                if (ready == 15u) break;
                if (prof) ++npass;
                if ((spins & 15) == 15) {
                    // (wave-uniform by construction -- and by readfirstlane for the compiler: a lane-divergent `dead` leaves a static
                    // path from the poll loads into the step body that skips the explicit wait below)
                    const int st_now = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (wall_clock64() - t_start > p.timeout_ticks || st_now != 0) {
                        dead = true;
                        break;
                    }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (!((ready >> g) & 1u)) ld[g] = __builtin_amdgcn_raw_buffer_load_b128(rs[par], voff, soff_w + g * 1024, 2);
                }
            }
            if (dead) break;
            // lane (half = lane >> 5, unit = lane & 31) holds rows 0 .. 3 of producers 8 w + 2 g + half: sum over g in registers
            f32x4 sacc;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sacc[r] = !tagged(r) ? ((__uint_as_float(ld[0][r]) + __uint_as_float(ld[1][r])) + __uint_as_float(ld[2][r])) + __uint_as_float(ld[3][r])
                                     : ((__uint_as_float(ld[0][r] & ~1u) + __uint_as_float(ld[1][r] & ~1u)) + __uint_as_float(ld[2][r] & ~1u)) +
                                           __uint_as_float(ld[3][r] & ~1u);
            *reinterpret_cast<f32x4*>(gsum + ((wave * 2 + (lane >> 5)) * UPC + (lane & 31)) * 4) = sacc;
        }
        if (n == 0) precompute();
        if (prof) st1 = wall_clock64();
        // every poll load has returned (its data was just compared) and, vmcnt retiring in order, so has everything issued before it;
        // say so in a form the compiler's wait-count pass sees ON EVERY PATH into the step body, or it protects the re-issued loads'
        // destination registers with a vmcnt(0) at their next reuse -- in the MFMA block, where it would wait for the ring DMAs and
        // output stores waves 2-3 have issued in between
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();                                             // (1) gather sums visible; dgates operands of the last step consumed
        if (n > 0) store_outputs(n - 1);
        prefetch(n + DIST);
        if (erole) {
            float da[4] = {0.f, 0.f, 0.f, 0.f};
            if (active) {
                float dh = dy_s;
                if (n > 0) {
                    const float* gs = gsum + el * 4 + ebl;
                    dh += ((gs[0] + gs[UPC * 4]) + (gs[2 * UPC * 4] + gs[3 * UPC * 4])) + ((gs[4 * UPC * 4] + gs[5 * UPC * 4]) + (gs[6 * UPC * 4] + gs[7 * UPC * 4]));
                }
                const float dc = __fmaf_rn(dh, fA, dc_carry);
                dc_carry = dc * fgate;
                da[0] = dc * fI; da[1] = dc * fF; da[2] = dc * fG; da[3] = dh * fO;
            }
            // dgates as MFMA A operands: daop[gate][row][unit pair] (16-bit pairs; the odd unit comes from lane + 1 of the DPP row)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float nb = __uint_as_float(row_shl<1>(__float_as_uint(da[g])));
                if ((el & 1) == 0) daop[(g * 16 + ebl) * (UPC / 2) + (el >> 1)] = pack_op16x2(da[g], nb);
            }
            float* o = outs + (n & 1) * 4 * NE + tid;
#pragma unroll
            for (int g = 0; g < 4; ++g) o[g * NE] = da[g];
        }
        __syncthreads();                                             // (2) the group's rows of dgates_s are in LDS
        if (prof) st2 = wall_clock64();
        if (n + 1 < tg) {
            // ---- partial dh_rec of the NEXT step: this CU's dgates x its 128 rows of W_hh; rows >= RPGP of the A tile are zero
            u32x4 a[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = *reinterpret_cast<const u32x4*>(daop + (g * 16 + li) * (UPC / 2) + kg * 4);
            // tiles in groups of four (16 MFMAs: each accumulator's chain is four instructions apart); a group's results are tagged and
            // stored while the NEXT group's MFMAs occupy the matrix pipe.  D rows 0 .. 3 sit in lanes kg == 0; tile j of wave w =
            // columns (16 w + j) 16 + li = consumer 8 w + (j >> 1), unit (j & 1) 16 + li: one tagged 16-byte store [unit][4 rows] per
            // tile, lane offset in the VGPR, consumer / half offsets scalar (raw buffer store: the compiler owns its data hazards)
            const unsigned tagw = (unsigned)(n >> 1) & 1u;
            const __amdgpu_buffer_rsrc_t wr = wrs[n & 1];
            // One MFMA and then ONE filler: the tag operations and the store of the PREVIOUS group's tiles are handed out one per MFMA gap
            // (a 16x16x32 MFMA occupies the pipe for ~17 cycles, about four issue slots: a single VALU / VMEM instruction in the gap is
            // free, a dozen of them in a row behind a group stall the pipe of a one-wave-per-SIMD kernel for their whole issue time).
            // Filler m = 4 f + s of a group: tile f of the previous group -- s = 0: tag word 0, s = 1: tag word 3 (v_and_or_b32 in
            // place: the accumulator is dead), s = 2: nothing, s = 3: the 16-byte store.  Tile f's last MFMA lies 4 + 3 f MFMAs back.
            auto tag_word = [&](f32x4& v, int r) {
                float x = v[r];
                asm volatile("v_and_or_b32 %0, %0, -2, %1" : "+v"(x) : "v"(tagw));
                v[r] = x;
            };
            auto store_tile = [&](int j, const f32x4& acc) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), wr, wvoff,
                                                       wsoff_w + (j >> 1) * (CPG * UPC * RPGP * 4) + (j & 1) * (16 * RPGP * 4), 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto filler = [&](int m, int ptq, f32x4 (&pacc)[4]) {
                const int f = m >> 2, sl = m & 3;
                if (sl == 0) { if (tagged(0)) tag_word(pacc[f], 0); if (NTAG == 4) tag_word(pacc[f], 1); }
                else if (sl == 1) { if (tagged(3)) tag_word(pacc[f], 3); if (NTAG == 4) tag_word(pacc[f], 2); }
                else if (sl == 3) store_tile(ptq * 4 + f, pacc[f]);     // (slot 3: behind the group's four first-MFMAs -- in slot 2 the allocator
                                                                         // handed the stored tile's registers to the very next MFMA's result)
            };
            auto group = [&](int tq, f32x4 (&acc)[4], int ptq, f32x4 (&pacc)[4]) {          // ptq < 0: no previous group to publish
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
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");           // (the last group's results: MFMA -> VALU read wait states)
#pragma unroll
            for (int m = 0; m < 16; ++m) filler(m, 3, accB);
        }
        if (prof) st3 = wall_clock64();
        if (prof && n < 1024) {
            long* o = p.prof + ((size_t)n * 4 + wave) * 5;
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = npass;
        }
    }
    if (dead) {
        if (lane == 0) atomicExch(p.status, 1);
        return;
    }
    __syncthreads();
    if (tg > 0) store_outputs(tg - 1);
    if (WF32 && ev)
        for (int t = tg; t < T; ++t) {
            float* dg = p.dgx + ((size_t)t * LB + eb) * 4 * PH + eu;
#pragma unroll
            for (int g = 0; g < 4; ++g) dg[(size_t)g * PH] = 0.f;
        }
    if constexpr (WIMG) {
        if (ovalid) {
            unsigned short* ip = p.dimg + (size_t)(ooff + olen) * p.dimg_ld + ou;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                atomicAdd(p.dbias + g * PH + ou, bsum[g]);
                ip[g * PH] = 0;
            }
        }
        int R = 0;
        for (int bb = 0; bb < B; ++bb) R += p.lens[bb] + 1;
        int Rz = (R + 32 + 255) & ~255;
        Rz = Rz < p.dimg_rows ? Rz : p.dimg_rows;
        for (int r = R + grp * CPG + q; r < Rz; r += NCU) {
            uint4* row = reinterpret_cast<uint4*>(p.dimg + (size_t)r * p.dimg_ld);
            for (int c = tid; c < 4 * PH / 8; c += 256) row[c] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

}  // namespace

static inline size_t al256p(size_t v) { return (v + 255) & ~size_t(255); }
#if FT_OPFMT == 0
long* ftint_persist_prof = nullptr;                  // shared with the fp16 build of this file
#else
extern long* ftint_persist_prof;
#endif
#define g_persist_prof ftint_persist_prof
// debug hook (scripts/exp/lstm_persist_bench.py): device buffer of 1024 x 4 x 5 int64 that the NEXT forward launches fill
// with per-step phase stamps of workgroup (group 0, slot 0); nullptr switches it off
#if FT_OPFMT == 0
extern "C" int ft_lstm_persist_debug_prof(void* dev_buf) { g_persist_prof = reinterpret_cast<long*>(dev_buf); return FT_OK; }

extern "C" int ft_lstm_persist_supported(int B, int H) {
    if (H != PH || B < 1 || B > 32) return 0;
    // CU count of the CURRENT device, cached per device ordinal (a process may drive several GPUs)
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    int n = dev < 64 ? __atomic_load_n(&cus[dev], __ATOMIC_RELAXED) : 0;
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (dev < 64) __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
    }
    return n >= NCU ? 1 : 0;
}

extern "C" size_t ft_lstm_persist_workspace_bytes(int B, int H) {
    (void)B;
    // W_hh fragment image + granule buffers (2 parities x 32 rows x K/2 granules x 8 B, independent of NG; K = H forward,
    // 4H backward) + census counters
    // (the reduce-scatter backward, transport 21, takes 2 parities x 8 groups x [32 x 32 x 32 x 4] fp32 partials = 8 MB instead)
    const size_t gran = (size_t)2 * 32 * (4 * H / 2) * 8, rs = (size_t)2 * 8 * 32 * 32 * 32 * 4 * sizeof(float);
    return al256p((size_t)4 * H * H * 2) + al256p(gran > rs ? gran : rs) + 256;
}
#endif

// ldb = batch rows per time step in memory (B, or the width of the batch this launch covers a slice of: ft_lstm_persist_fwd_rows)
static int persist_fwd_impl(const float* gx, const float* w_hh, const int32_t* lens, float* y, int64_t ldy,
                            float* gates, float* cell, void* work, int32_t* status, int T, int B, int ldb, int H, int ng,
                            void* stream) {
    FT_CHECK_ARG(gx && w_hh && lens && y && work && status && ldb >= B);
    FT_CHECK_ARG((gates == nullptr) == (cell == nullptr));
    FT_CHECK_ARG(T >= 0 && ldy >= H && reinterpret_cast<uintptr_t>(work) % 256 == 0 && reinterpret_cast<uintptr_t>(gx) % 16 == 0);
    // ng: 1 / 9 = XCD-local transport (nt / sc1 loads) with tagged granules; 11 / 19 = the same with BARE operand pairs and the sentinel
    // protocol.  (Round 5 pruned what no box has run since round 3: the placement-independent fabric transports 8 | 4 | 2 | 18 | 14 | 12
    // -- the kernel templates still carry their LOCAL = false branches -- and the M-split kernel of transport 31, a measured loser.)
    const bool bare = ng > 10;
    const int ngb = bare ? ng - 10 : ng;
    FT_CHECK_ARG(ngb == 1 || ngb == 9);
    if (!ft_lstm_persist_supported(B, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_lstm_persist_fwd: needs H == 1024, B <= 32 and a 256-CU device (H=%d B=%d)", H, B);
    if (T == 0) return FT_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(work);
    unsigned short* wfrag = reinterpret_cast<unsigned short*>(base);
    unsigned long long* hgran = reinterpret_cast<unsigned long long*>(base + al256p((size_t)4 * H * H * 2));
    // tagged: 2 parities x 32 rows x H/2 granules x 8 B; bare: 3 buffers x 32 rows x H/2 dwords (smaller)
    const size_t gran_bytes = al256p((size_t)2 * 32 * (H / 2) * 8);
    unsigned* census = reinterpret_cast<unsigned*>(base + al256p((size_t)4 * H * H * 2) + al256p((size_t)2 * 32 * (4 * H / 2) * 8));
    // tags = 0: no epoch matches (epochs start at 1); bare: sentinels.  Preset by the fragment kernel (lstm_images.h: WfragAux)
    const WfragAux aux{reinterpret_cast<uint4*>(hgran), (unsigned long)(gran_bytes / 16), bare ? 0xFFFFFFFFu : 0u, census};
    hipLaunchKernelGGL(make_wfrag_fwd_ug, dim3(2048), dim3(256), 0, st, w_hh, wfrag, H, aux);
    PersistP p{gx, lens, y, (long)ldy, gates, cell, wfrag, hgran, status, census, T, B, 100000000L / 2, g_persist_prof, ldb};   // 0.5 s
    // dynamic LDS: reduce buffers (2*4*TPC*RPGP*20 = 2*4*32*20 floats) + SB staged gx rows + 2 output rows
    const size_t lds = sizeof(float) * ((size_t)2 * 4 * 32 * 20 + (size_t)SB * 4 * 128 + (size_t)2 * 6 * 128);
    auto launch = [&](auto kern) -> int {
        FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(NCU), dim3(256), lds, st, p);
        return FT_OK;
    };
    int rc;
    if (!bare) rc = ngb == 1 ? launch(lstm_persist_fwd_k<8, true, 2, false>) : launch(lstm_persist_fwd_k<8, true, 16, false>);
    else rc = ngb == 1 ? launch(lstm_persist_fwd_k<8, true, 2, true>) : launch(lstm_persist_fwd_k<8, true, 16, true>);
    if (rc != FT_OK) return rc;
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_lstm_persist_bwd_img)(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                                   const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng,
                                   void* dimg, int64_t dimg_ld, int64_t dimg_rows, float* dbias, void* stream);

extern "C" int FT_OPNAME(ft_lstm_persist_fwd)(const float* gx, const float* w_hh, const int32_t* lens, float* y, int64_t ldy,
                                   float* gates, float* cell, void* work, int32_t* status, int T, int B, int H, int ng,
                                   void* stream) {
    return persist_fwd_impl(gx, w_hh, lens, y, ldy, gates, cell, work, status, T, B, B, H, ng, stream);
}
extern "C" int FT_OPNAME(ft_lstm_persist_fwd_rows)(const float* gx, const float* w_hh, const int32_t* lens, float* y, int64_t ldy,
                                   float* gates, float* cell, void* work, int32_t* status, int T, int B, int ldb, int H, int ng,
                                   void* stream) {
    return persist_fwd_impl(gx, w_hh, lens, y, ldy, gates, cell, work, status, T, B, ldb, H, ng, stream);
}

static int persist_bwd_impl(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                            const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int ldb, int H, int ng,
                            void* dimg, int64_t dimg_ld, int64_t dimg_rows, float* dbias, void* stream);

extern "C" int FT_OPNAME(ft_lstm_persist_bwd)(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                                   const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng,
                                   void* stream) {
    return persist_bwd_impl(dy, ldy, w_hh, lens, gates, cell, dgx, work, status, T, B, B, H, ng, nullptr, 0, 0, nullptr, stream);
}
extern "C" int FT_OPNAME(ft_lstm_persist_bwd_rows)(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                                   const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int ldb, int H, int ng,
                                   void* stream) {
    return persist_bwd_impl(dy, ldy, w_hh, lens, gates, cell, dgx, work, status, T, B, ldb, H, ng, nullptr, 0, 0, nullptr, stream);
}

extern "C" int FT_OPNAME(ft_lstm_persist_bwd_img)(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                                   const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng,
                                   void* dimg, int64_t dimg_ld, int64_t dimg_rows, float* dbias, void* stream) {
    return persist_bwd_impl(dy, ldy, w_hh, lens, gates, cell, dgx, work, status, T, B, B, H, ng, dimg, dimg_ld, dimg_rows, dbias, stream);
}

static int persist_bwd_impl(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                            const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int ldb, int H, int ng,
                            void* dimg, int64_t dimg_ld, int64_t dimg_rows, float* dbias, void* stream) {
    FT_CHECK_ARG(dy && w_hh && lens && gates && cell && (dgx || dimg) && work && status && ldb >= B && (dimg == nullptr || ldb == B));
    FT_CHECK_ARG(dimg == nullptr || (dbias && dimg_ld >= 4 * (int64_t)H && dimg_ld % 8 == 0 && dimg_rows >= (int64_t)T * B + B &&
                                     reinterpret_cast<uintptr_t>(dimg) % 16 == 0));
    FT_CHECK_ARG(T >= 0 && ldy >= H && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    // ng = 21: the reduce-scatter form (lstm_persist_bwd_rs_k: XCD-local, fp32 partials tagged in the mantissa LSB)
    const bool rsform = ng == 21;
    const bool bare = ng > 10 && !rsform;
    const int ngb = rsform ? 1 : (bare ? ng - 10 : ng);
    FT_CHECK_ARG(ngb == 1 || ngb == 9);
    if (!ft_lstm_persist_supported(B, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_lstm_persist_bwd: needs H == 1024, B <= 32 and a 256-CU device (H=%d B=%d)", H, B);
    if (T == 0) return FT_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(work);
    unsigned short* wTfrag = reinterpret_cast<unsigned short*>(base);
    unsigned long long* dgran = reinterpret_cast<unsigned long long*>(base + al256p((size_t)4 * H * H * 2));
    const size_t rs_bytes = (size_t)2 * 8 * 32 * 32 * 32 * 4 * sizeof(float);
    const size_t gran_bytes = al256p(rsform ? rs_bytes : (size_t)2 * 32 * (4 * H / 2) * 8);
    unsigned* census = reinterpret_cast<unsigned*>(base + ft_lstm_persist_workspace_bytes(B, H) - 256);
    const WfragAux aux{reinterpret_cast<uint4*>(dgran), (unsigned long)(gran_bytes / 16), (bare || rsform) ? 0xFFFFFFFFu : 0u, census};
    if (rsform) hipLaunchKernelGGL(make_wfrag_rs, dim3(2048), dim3(256), 0, st, w_hh, wTfrag, H, aux);
    else hipLaunchKernelGGL(make_wfrag_bwd, dim3(2048), dim3(256), 0, st, w_hh, wTfrag, H, aux);
    PersistBwdP p{dy, (long)ldy, lens, gates, cell, dgx, wTfrag, dgran, status, census, T, B, 100000000L / 2, g_persist_prof,
                  reinterpret_cast<unsigned short*>(dimg), (long)dimg_ld, (int)dimg_rows, dbias, ldb};
    // dynamic LDS: 16-partial reduce (2 x 16 x 32 unit-rows... = 2*16*TL*RPGP*17 = 2*16*8*17 floats) + staged steps
    const size_t lds = sizeof(float) * ((size_t)2 * 16 * 8 * 17 + (size_t)RING * 5 * 128 + (size_t)RING * 128 + (size_t)2 * 4 * 128);
    auto launch = [&](auto kern) -> int {
        FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(NCU), dim3(256), lds, st, p);
        return FT_OK;
    };
    int rc;
    const int out = dimg == nullptr ? 0 : (dgx ? 1 : 2);
    if (rsform) {
        // (the ring is filled by 16-byte-per-lane LDS-DMA pieces: rows of the saved tensors and of dy 16-byte aligned)
        FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dy) % 16 == 0 && ldy % 4 == 0 && reinterpret_cast<uintptr_t>(gates) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(cell) % 16 == 0);
        // LDS: 8 x 128 gather sums + 1024 dwords of dgates operands (16-row A tiles) + the ring (5 + 1 rows per slot) + 2 output rows
        const size_t lds_rs = sizeof(float) * ((size_t)8 * 128 + 1024 + (size_t)RING * 5 * 128 + (size_t)RING * 128 + (size_t)2 * 4 * 128);
        auto launch_rs = [&](auto kern) -> int {
            FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rs));
            hipLaunchKernelGGL(kern, dim3(NCU), dim3(256), lds_rs, st, p);
            return FT_OK;
        };
        if (g_persist_prof) rc = launch_rs(lstm_persist_bwd_rs_k<0, true>);      // debug stamps: fp32 rows, whatever was asked for
        else rc = out == 0 ? launch_rs(lstm_persist_bwd_rs_k<0>) : out == 1 ? launch_rs(lstm_persist_bwd_rs_k<1>) : launch_rs(lstm_persist_bwd_rs_k<2>);
        if (rc != FT_OK) return rc;
        FT_CHECK_LAUNCH();
        return FT_OK;
    }
#define FT_PBWD(NG_, LOCAL_, LAUX_, BARE_) \
    (out == 0 ? launch(lstm_persist_bwd_k<NG_, LOCAL_, LAUX_, BARE_, 0>) : out == 1 ? launch(lstm_persist_bwd_k<NG_, LOCAL_, LAUX_, BARE_, 1>) \
                                                                                  : launch(lstm_persist_bwd_k<NG_, LOCAL_, LAUX_, BARE_, 2>))
    if (!bare) rc = ngb == 1 ? FT_PBWD(8, true, 2, false) : FT_PBWD(8, true, 16, false);
    else rc = ngb == 1 ? FT_PBWD(8, true, 2, true) : FT_PBWD(8, true, 16, true);
#undef FT_PBWD
    if (rc != FT_OK) return rc;
    FT_CHECK_LAUNCH();
    return FT_OK;
}
