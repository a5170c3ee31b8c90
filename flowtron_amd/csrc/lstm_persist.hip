// Persistent LSTM recurrence, BACKWARD, reduce-scatter form (round 4): ONE launch for the whole sequence, W_hh resident in the
// accumulation registers, 8 batch groups = the 8 XCDs (run-time XCC census), 4 batch rows each, hand-offs through the XCD's own L2.
//
// History of this file: rounds 2-3 built the forward kernel (lstm_persist_fwd_k: all-gather of h as tagged
// granules, later bare operand pairs behind a sentinel) and the all-gather backward (lstm_persist_bwd_k); round 4 the reduce-scatter
// backward below, which has been the step's backward recurrence since.  Round 6 removed what it superseded: the forward recurrence
// lives in lstm_roles.hip (rows per XCD group / windows / roles; 1.62 us per step at 4 rows against 1.76 for lstm_persist_fwd_k,
// bit-identical), the all-gather backward (2.84 us per step against 1.62), the tagged-granule transport and the placement-
// independent (fabric) template branches are gone.  Every spin is bounded by a wall-clock timeout that raises status[0]; the host
// checks it (a chip with fewer than 256 free CUs cannot host the grid) and falls back to the launch-per-step kernels of lstm.hip.
#include "lstm_persist_common.h"

namespace {

constexpr int RING = 8, DIST = 6;        // LDS ring of staged steps: a slot is filled DIST steps ahead by LDS-DMA from waves 2-3
static_assert(DIST < RING - 1, "slot reuse");

struct PersistBwdP {
    const float* dy; long ldy; const int* lens;
    const float* gates; const float* cell; float* dgx;
    const unsigned short* wTfrag;        // make_wfrag_rs image
    unsigned long long* dgran;           // partial buffers [2 parity][8 groups][consumer 32][producer 32][32 units][4 rows] fp32
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

    // output role of waves 2-3: thread tid >= 128 stores the dgates row / image entries of the previous step
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
// debug hook (scripts/exp/lstm_roles_bench.py): device buffer of 1024 x 4 x 5 int64 that the NEXT backward launches fill with per-step
// phase stamps of workgroup (group 0, slot 0); nullptr switches it off
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
    // W_hh fragment image + the partial buffers of the reduce-scatter hand-off (2 parities x 8 groups x [32 x 32 x 32 x 4] fp32 = 8 MB)
    // + census counters
    return al256p((size_t)4 * H * H * 2) + al256p((size_t)2 * 8 * 32 * 32 * 32 * 4 * sizeof(float)) + 256;
}
#endif

// ldb = batch rows per time step in memory (B here; the kernel takes the stride separately)
static int persist_bwd_impl(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                            const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int ldb, int H, int ng,
                            void* dimg, int64_t dimg_ld, int64_t dimg_rows, float* dbias, void* stream) {
    FT_CHECK_ARG(dy && w_hh && lens && gates && cell && (dgx || dimg) && work && status && ldb >= B && (dimg == nullptr || ldb == B));
    FT_CHECK_ARG(dimg == nullptr || (dbias && dimg_ld >= 4 * (int64_t)H && dimg_ld % 8 == 0 && dimg_rows >= (int64_t)T * B + B &&
                                     reinterpret_cast<uintptr_t>(dimg) % 16 == 0));
    FT_CHECK_ARG(T >= 0 && ldy >= H && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    // ng: 21 = the reduce-scatter form (lstm_persist_bwd_rs_k: XCD-local, fp32 partials tagged in the mantissa LSB) -- the one
    // transport left (round 6 removed the all-gather kernels 1 / 9 / 11 / 19); 1 is accepted as "the default"
    FT_CHECK_ARG(ng == 21 || ng == 1);
    if (!ft_lstm_persist_supported(B, H))
        return ft_fail(FT_EUNSUPPORTED, "ft_lstm_persist_bwd: needs H == 1024, B <= 32 and a 256-CU device (H=%d B=%d)", H, B);
    if (T == 0) return FT_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(work);
    unsigned short* wTfrag = reinterpret_cast<unsigned short*>(base);
    unsigned long long* dgran = reinterpret_cast<unsigned long long*>(base + al256p((size_t)4 * H * H * 2));
    const size_t gran_bytes = al256p((size_t)2 * 8 * 32 * 32 * 32 * 4 * sizeof(float));
    unsigned* census = reinterpret_cast<unsigned*>(base + ft_lstm_persist_workspace_bytes(B, H) - 256);
    // hand-off preset (0xFF: tag 1 != the tag of steps 0 / 1) and census zero ride on the fragment kernel (lstm_images.h: WfragAux)
    const WfragAux aux{reinterpret_cast<uint4*>(dgran), (unsigned long)(gran_bytes / 16), 0xFFFFFFFFu, census};
    hipLaunchKernelGGL(make_wfrag_rs, dim3(2048), dim3(256), 0, st, w_hh, wTfrag, H, aux);
    PersistBwdP p{dy, (long)ldy, lens, gates, cell, dgx, wTfrag, dgran, status, census, T, B, 100000000L / 2, g_persist_prof,
                  reinterpret_cast<unsigned short*>(dimg), (long)dimg_ld, (int)dimg_rows, dbias, ldb};
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
    const int out = dimg == nullptr ? 0 : (dgx ? 1 : 2);
    int rc;
    if (g_persist_prof) rc = launch_rs(lstm_persist_bwd_rs_k<0, true>);      // debug stamps: fp32 rows, whatever was asked for
    else rc = out == 0 ? launch_rs(lstm_persist_bwd_rs_k<0>) : out == 1 ? launch_rs(lstm_persist_bwd_rs_k<1>) : launch_rs(lstm_persist_bwd_rs_k<2>);
    if (rc != FT_OK) return rc;
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_lstm_persist_bwd)(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                                   const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng,
                                   void* stream) {
    return persist_bwd_impl(dy, ldy, w_hh, lens, gates, cell, dgx, work, status, T, B, B, H, ng, nullptr, 0, 0, nullptr, stream);
}

extern "C" int FT_OPNAME(ft_lstm_persist_bwd_img)(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                                   const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng,
                                   void* dimg, int64_t dimg_ld, int64_t dimg_rows, float* dbias, void* stream) {
    return persist_bwd_impl(dy, ldy, w_hh, lens, gates, cell, dgx, work, status, T, B, B, H, ng, dimg, dimg_ld, dimg_rows, dbias, stream);
}
