// hr_kernels_fused.hip -- the hot kernel: K-weighting + hop energies + true peak in ONE read of the PCM.
// Stereo f32 (or s16, or packed s24) interleaved input; the true peak of either bank (libswresample's to 192 kHz,
// BS.1770 Annex 2).
//
// Decomposition.  A wave owns up to 64 consecutive rows of one track, one row per lane; a row is a hop (100 ms) or a
// whole fraction of one (FusedItem::row_len: the host picks it per launch and track, run_measure).  The 64 rows advance
// in lock step, 16 frames (one 128-byte piece per row) per step.
//   * HBM -> LDS: eight global_load_lds_dwordx4 per step move the 64 pieces (8 KiB) with every 128-byte
//     piece read by 8 adjacent lanes (full cache lines), no VGPR staging.  The LDS image is lane-linear, so
//     the bank swizzle is applied on the SOURCE chunk index (chunk ^ ((row>>1)&7)) and undone by the reader:
//     each lane then reads its own row with eight conflict-free ds_read_b128.
//   * K1/K2a (VALU, f64): every hop is filtered from ZERO state (input history taken from the data), and
//     records energy e0, four correction dot products q and its end state; k_hop_combine later applies the
//     exact carry-in state (H = e0 + 2 s.q + s'Gs).  Same operations in the same order as
//     k_hop_zero_state, so the records are bit-identical to the generic kernel's.
//   * K3 (MFMA, f16 screen + exact f64 refinement): the 16 positions of a tile and their 4.35 (44.1 kHz) or
//     4 (48 kHz) interpolated outputs each are a [<=96 outputs] x [48-frame window] coefficient tile times a
//     [48] x [64 rows] signal matrix: v_mfma_f32_32x32x16_f16 with the rows as the N dimension.  The B
//     fragments come straight from the lane's own registers (v_cvt_pkrtz + v_permlane32_swap), never
//     from LDS.  The f16 product is only a SCREEN with a rigorous error bound err = 4.1e-3*S + 4e-7
//     (S = sample peak seen so far): an output can be the maximum only if |approx| + err >= lb, where
//     lb = max(S, max|approx| - err) is a proven lower bound of the true peak.  Those few candidates (in
//     groups of the 16 outputs a lane holds of one accumulator) are appended to a per-wave list (no
//     atomics, no waiting) and k_refine_cands re-evaluates the ones that
//     still matter against the track's final bound exactly (f64 FMA chain in the oracle's tap order), so
//     the reported peak is the exact f64 maximum over all outputs: bit-identical to the generic kernel and
//     to the oracle.  A wave whose list overflows flags its track, which then takes the generic kernel.
//
// Rows cover the whole track where the stream ends with it: the final hop and the ragged tail behind it are rows too,
// kept honest behind wave-uniform branches in the last steps of the one wave that holds them ("the track's LAST rows"
// below); the few positions behind the last screened tile go through the bounds-checked generic kernel.  Everywhere
// else this kernel needs no bounds checks at all.
//
// Replaces FFmpeg f_ebur128.c FILTER()/i400 accumulation and swr_convert + fabs max (reached from
// src/analyzer.rs:253-267); SURVEY.md Appendix A.2 / A.4.
#include "hr_fused_core.hpp"

namespace hr {

#ifdef HR_EXPERIMENTS
// ---- k_fused_split: the same pass with the two halves of the work in SEPARATE waves -------------------------------
// A workgroup is eight waves = four pairs; a pair owns one item (64 rows).  Wave p (0..3) of a pair K-weights
// (f64 VALU only), wave p+4 screens the true peak (f16 MFMA + f32 VALU); both read the SAME 8 KiB LDS image of the
// piece, which each of them half fills by LDS-DMA, so the PCM is still read from HBM once.  The piece buffer is
// in a ring of four buffers (pieces k+1 .. k+3 are in flight while piece k is consumed), so one s_barrier per step is enough.  The two
// instruction streams are independent: the SIMD's own arbiter overlaps the f64 chain of one wave with the matrix
// work of the other, instead of a hand-interleaved schedule inside one wave.  Arithmetic (operation order of the
// K-weighting, screen bounds, candidate lists) is exactly that of k_fused_f32s: the records and peaks are
// bit-identical.
// Measured (profiles/README.md, round 1): NOT faster -- 4.70-4.79 ms against 4.52 ms for 200 tracks, with one
// 148 KiB workgroup per CU; deeper piece rings change nothing.  It is therefore opt-in (HR_FUSED_SPLIT=1) and kept
// as the vehicle for per-role timing: wave_map bit 0 = pairing of waves (0: w / w+4, 1: 2p / 2p+1), bit 1 = the
// K-weighting waves only move data, bit 2 = the true-peak waves only move data (results are wrong with bits 1, 2).
constexpr int split_lds(int pairs, int ring) { return pairs * ring * kPieceLds + 2 * kCoefLds + 2 * kCarryLds; }

// PAIRS: items per workgroup (2 or 4); RING: piece buffers per pair (a power of two; RING-1 pieces in flight)
template <int MT, bool SMALL, int PAIRS, int RING>
__global__ void __launch_bounds__(PAIRS * 128, PAIRS == 2 ? 3 : 2) k_fused_split(const TrackDesc *__restrict__ tracks,
                                                     const RateDev *__restrict__ rates,
                                                     const FusedItem *__restrict__ items, HopRec *__restrict__ recs,
                                                     uint2 *__restrict__ cands, unsigned *__restrict__ cand_cnt,
                                                     unsigned *__restrict__ tpb, unsigned *__restrict__ overflow,
                                                     unsigned long long *__restrict__ peaks, int wave_map)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];  // split_lds(PAIRS, RING)
    constexpr int D = RING - 1;  // prefetch distance in pieces
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = (wave_map & 1) ? wave >> 1 : wave % PAIRS;
    // wave_map bit 3: swap the roles in every other workgroup, so that a SIMD that always receives the same wave
    // index of every workgroup still gets both kinds of work
    const bool swap = (wave_map & 8) && (((blockIdx.x >> 8) ^ (blockIdx.x >> 3) ^ blockIdx.x) & 1);
    const bool tp_role = ((wave_map & 1) ? (wave & 1) != 0 : wave >= PAIRS) != swap;
    const uint32_t item_idx = blockIdx.x * PAIRS + pair;
    const FusedItem it = items[item_idx];
    const TrackDesc &t = tracks[it.track];
    const RateDev &r = rates[t.rate_idx];
    const bool passthru0 = r.tp_passthru0 != 0;
    const uint32_t hop = t.hop;
    const int T = static_cast<int>((hop + 15) / 16);
    const int n_rows = static_cast<int>(it.n_rows);
    const int last_row = n_rows > 0 ? n_rows - 1 : 0;
    const int row = lane < n_rows ? lane : last_row;
    const uint64_t h = it.first_hop + row;
    const char *wave_base = static_cast<const char *>(t.pcm) + it.first_hop * hop * 8ull;
    char *pbuf = smem + pair * RING * kPieceLds;   // [RING][8 KiB]: piece k lives in buffer k mod RING
    char *abuf = smem + PAIRS * RING * kPieceLds;  // [2][9 KiB]
    char *gbuf = abuf + 2 * kCoefLds;          // [2][1 KiB]
    const char *gsrc = reinterpret_cast<const char *>(r.g);
    const unsigned lane16 = static_cast<unsigned>(lane) * 16u;
    const int T_full = static_cast<int>(hop / 16), rem = static_cast<int>(hop % 16);
    const int k_short = r.q_n0 < hop ? static_cast<int>(r.q_n0 / 16) : T + 1;

    // this wave's half of the piece DMA: instructions jbase .. jbase+3 of the eight (slots 8j .. 8j+7 each)
    const int jbase = tp_role ? 4 : 0;
    unsigned dma_off[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int slot = 8 * (jbase + j) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((slot >> 1) & 7);
        const int srow = slot < n_rows ? slot : last_row;
        dma_off[j] = static_cast<unsigned>(srow) * hop * 8u + static_cast<unsigned>(chunk) * 16u;
    }
    const unsigned rd_swz = (lane >> 1) & 7;
    auto issue_piece = [&](int k) {
        const char *src = wave_base + static_cast<size_t>(k) * 128;
        char *dst = pbuf + (k & (RING - 1)) * kPieceLds + jbase * 1024;
#pragma unroll
        for (int j = 0; j < 4; j++) dma16(src, dma_off[j], dst + j * 1024);
    };

    // The piece is read with hidden LDS reads too: piece k+1 is still in flight when piece k is consumed, and the
    // compiler would wait for every outstanding DMA in front of an LDS read it can see.
    unsigned rd_off[8];
#pragma unroll
    for (int c = 0; c < 8; c++) rd_off[c] = lds_addr(pbuf) + static_cast<unsigned>(lane) * 128u + ((static_cast<unsigned>(c) ^ rd_swz) * 16u);
    float4 v[8];
    auto read_piece = [&](int k) {
        uint4v u[8];
#pragma unroll
        for (int c = 0; c < 8; c++) u[c] = lds_ld128(rd_off[c] + static_cast<unsigned>(k & (RING - 1)) * kPieceLds, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : : "memory");
#pragma unroll
        for (int c = 0; c < 8; c++)
            v[c] = make_float4(__uint_as_float(u[c][0]), __uint_as_float(u[c][1]), __uint_as_float(u[c][2]), __uint_as_float(u[c][3]));
    };
    // everything this wave has in flight except the newest piece (its four DMAs are always the last ones issued)
    auto wait_step = [&](int k) {
        if (D >= 3 && k + 2 <= T) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (D >= 2 && k < T) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    // ---- prologue: the 16 frames in front of the row
    {
        const float *rowp = reinterpret_cast<const float *>(wave_base + static_cast<size_t>(row) * hop * 8u);
        if (h > 0) {
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = *reinterpret_cast<const float4 *>(rowp - 32 + 4 * c);
        } else if (t.left_tail) {
#pragma unroll
            for (int c = 0; c < 8; c++)
                v[c] = make_float4(static_cast<float>(t.left_tail[(2 * c) * 2 + 0]), static_cast<float>(t.left_tail[(2 * c) * 2 + 1]),
                                   static_cast<float>(t.left_tail[(2 * c + 1) * 2 + 0]), static_cast<float>(t.left_tail[(2 * c + 1) * 2 + 1]));
        } else if (t.tp_reflect) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float2 a = *reinterpret_cast<const float2 *>(rowp + 2 * (16 - 2 * c));
                const float2 b = *reinterpret_cast<const float2 *>(rowp + 2 * (15 - 2 * c));
                v[c] = make_float4(a.x, a.y, b.x, b.y);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    const bool have_history = h > 0 || t.left_tail;  // else: start of the stream, the K-weighting starts from silence

    if (!tp_role) {
        // ================================ K-weighting wave ================================
        const KCoef kc = r.k;
        KState ks[2];
#pragma unroll
        for (int c = 0; c < 2; c++) ks[c] = KState{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (have_history) {
            ks[0].x1 = v[7].z; ks[1].x1 = v[7].w;
            ks[0].x2 = v[7].x; ks[1].x2 = v[7].y;
        }
        if (pair == 0) dma16(gsrc, lane16, gbuf);
#pragma unroll
        for (int d = 0; d < D; d++) issue_piece(d);

        auto kw_step = [&](int k, auto short_c) {
            constexpr bool SHORT = decltype(short_c)::value;
            wait_step(k);
            __builtin_amdgcn_s_barrier();
            const char *my_lds = pbuf + (k & (RING - 1)) * kPieceLds + lane * 128;
            read_piece(k);
            if (rem && k == T_full) {
                const char *gk = gbuf + (k & 1) * kCarryLds;
                for (int i = 0; i < rem; i++) {
                    const float2 x = *reinterpret_cast<const float2 *>(my_lds + (((i >> 1) ^ rd_swz) * 16) + (i & 1) * 8);
                    if constexpr (!SHORT) {
                        const G4 g4 = lds_g4(gk + i * 32);
                        kw_sample(ks[0], kc, x.x, g4);
                        kw_sample(ks[1], kc, x.y, g4);
                    } else {
                        const double2 g = *reinterpret_cast<const double2 *>(gk + i * 32 + 16);
                        kw_sample_short(ks[0], kc, x.x, g.x, g.y);
                        kw_sample_short(ks[1], kc, x.y, g.x, g.y);
                    }
                }
            }
            if (k < T && pair == 0) dma16(gsrc + static_cast<size_t>(k + 1) * 512, lane16, gbuf + ((k + 1) & 1) * kCarryLds);
            if (k + D <= T) issue_piece(k + D);
            if (k < T_full && !(wave_map & 2)) {
                const unsigned g_at = lds_addr(gbuf) + static_cast<unsigned>(k & 1) * kCarryLds;  // broadcast reads
                struct GC { uint4v r[4]; };
                auto load_gc = [&](int c) {
                    GC o;
                    if constexpr (!SHORT) {
                        o.r[0] = lds_ld128(g_at, (2 * c) * 32);
                        o.r[1] = lds_ld128(g_at, (2 * c) * 32 + 16);
                        o.r[2] = lds_ld128(g_at, (2 * c + 1) * 32);
                        o.r[3] = lds_ld128(g_at, (2 * c + 1) * 32 + 16);
                    } else {
                        o.r[1] = lds_ld128(g_at, (2 * c) * 32 + 16);
                        o.r[3] = lds_ld128(g_at, (2 * c + 1) * 32 + 16);
                        o.r[0] = o.r[2] = uint4v{0, 0, 0, 0};
                    }
                    return o;
                };
                auto wait_gc = [&](GC &g) {
                    if constexpr (!SHORT) lds_wait(g.r[0], g.r[1], g.r[2], g.r[3]);
                    else lds_wait(g.r[1], g.r[3]);
                };
                GC gcur = load_gc(0);
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    wait_gc(gcur);
                    GC gnext = gcur;
                    if (c + 1 < 8) gnext = load_gc(c + 1);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float x = i == 0 ? v[c].x : i == 1 ? v[c].y : i == 2 ? v[c].z : v[c].w;
                        const double2 g23 = as_double2(gcur.r[i < 2 ? 1 : 3]);
                        if constexpr (!SHORT) {
                            const double2 g01 = as_double2(gcur.r[i < 2 ? 0 : 2]);
                            kw_sample(ks[i & 1], kc, x, G4{g01.x, g01.y, g23.x, g23.y});
                        } else {
                            kw_sample_short(ks[i & 1], kc, x, g23.x, g23.y);
                        }
                    }
                    gcur = gnext;
                }
            }
        };
        const int k_split = k_short < T + 1 ? k_short : T + 1;
        for (int k = 0; k < k_split; k++) kw_step(k, std::false_type{});
        if (k_split <= T) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                ks[c].q0 -= fma(r.q_ab[0][0], ks[c].q2, r.q_ab[0][1] * ks[c].q3);
                ks[c].q1 -= fma(r.q_ab[1][0], ks[c].q2, r.q_ab[1][1] * ks[c].q3);
            }
        }
        for (int k = k_split; k < T + 1; k++) kw_step(k, std::true_type{});
        if (lane < n_rows) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (k_short <= T) {
                    ks[c].q0 += fma(r.q_ab[0][0], ks[c].q2, r.q_ab[0][1] * ks[c].q3);
                    ks[c].q1 += fma(r.q_ab[1][0], ks[c].q2, r.q_ab[1][1] * ks[c].q3);
                }
                HopRec &o = recs[t.rec_base + (kPreRec + h) * 2 + c];
                o.e0 = ks[c].e0;
                o.q[0] = ks[c].q0; o.q[1] = ks[c].q1; o.q[2] = ks[c].q2; o.q[3] = ks[c].q3;
                o.s[0] = ks[c].y1; o.s[1] = ks[c].y1 - ks[c].y2; o.s[2] = ks[c].z1; o.s[3] = ks[c].z1 - ks[c].z2;
            }
        }
        return;
    }

    // ================================ true-peak wave ================================
    // the coefficient area starts as zeros: step 0 screens a tile that does not exist and must find nothing
#pragma unroll
    for (int i = 0; i < 3 * MT; i++) *reinterpret_cast<uint4v *>(abuf + i * 1024 + lane * 16) = uint4v{0, 0, 0, 0};

    uint4v hist[3][2][2];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int n = 0; n < 2; n++) hist[a][c][n] = uint4v{0, 0, 0, 0};
    float s_lane[2] = {0.0f, 0.0f}, alb[2] = {0.0f, 0.0f};
    float S_u[2] = {-1.0f, -1.0f}, A_u[2] = {0.0f, 0.0f}, lb[2] = {0.0f, 0.0f};
    float thr_l[2][2];
#pragma unroll
    for (int c = 0; c < 2; c++) thr_l[c][0] = thr_l[c][1] = __builtin_inff();
    unsigned n_cand = 0;
    uint2 *__restrict__ my_cands = cands + static_cast<size_t>(item_idx) * kCandCap;

    // wave-wide bounds and this lane's candidate thresholds from the per-lane maxima seen so far
    auto refresh_bounds = [&]() {
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            S_u[ch] = wave_max_f32(s_lane[ch]);
            A_u[ch] = wave_max_f32(alb[ch]);
            const float err = fmaf(kScreenRel, S_u[ch], kScreenAbs);
            lb[ch] = fmaxf(lb[ch], fmaxf(passthru0 ? S_u[ch] : 0.0f, A_u[ch] - err));
            const float thr = fmaxf(lb[ch] - err, 1e-30f);  // silence is never a candidate; |x| beyond f16 -> overflow flag
            thr_l[ch][0] = (lane & 31) < n_rows ? thr : __builtin_inff();
            thr_l[ch][1] = 32 + (lane & 31) < n_rows ? thr : __builtin_inff();
        }
    };
    auto make_frags = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            s_lane[0] = fmaxf(fmaxf(fabsf(v[c].x), fabsf(v[c].z)), s_lane[0]);
            s_lane[1] = fmaxf(fmaxf(fabsf(v[c].y), fabsf(v[c].w)), s_lane[1]);
        }
        unsigned f[2][8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            f[0][c] = pk_f16(v[c].x, v[c].z);
            f[1][c] = pk_f16(v[c].y, v[c].w);
        }
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const auto sw = __builtin_amdgcn_permlane32_swap(f[ch][j], f[ch][j + 4], false, false);
                hist[slot][ch][0][j] = sw[0];
                hist[slot][ch][1][j] = sw[1];
            }
    };
    make_frags(std::integral_constant<int, 2>{});  // the prologue piece is "piece -1"

    const int NP = r.tp_np;
    int pat = 0;
    // this wave's share (chunks i with i mod 4 == pair) of the coefficients of the tile that step k+1 screens
    auto issue_coeff = [&](int pattern, int k, int kmod3) {
        const char *asrc = reinterpret_cast<const char *>(r.tp_atab) + static_cast<size_t>(pattern) * MT * 3 * 1024;
        char *dst = abuf + ((k + 1) & 1) * kCoefLds;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int ksx = 0; ksx < 3; ksx++)
                if ((mt * 3 + ksx) % PAIRS == pair)
                    dma16(asrc + (mt * 3 + ksx) * 1024, lane16, dst + (mt * 3 + (kmod3 + 2 + ksx) % 3) * 1024);
    };
#pragma unroll
    for (int d = 0; d < D; d++) issue_piece(d);

    auto tp_step = [&](int k, auto slot_c) {
        constexpr int KK = decltype(slot_c)::value;  // k mod 3
        wait_step(k);
        __builtin_amdgcn_s_barrier();
        read_piece(k);
        if (k < T) {
            issue_coeff(pat, k, KK);
            pat = pat + 1 == NP ? 0 : pat + 1;
        }
        if (k + D <= T) issue_piece(k + D);
        if (wave_map & 4) return;
        make_frags(slot_c);
        {
            const float over = fmaxf(fmaxf(s_lane[0] - S_u[0], s_lane[1] - S_u[1]), fmaxf(alb[0] - A_u[0], alb[1] - A_u[1]));
            const bool stale = over > 0.0f;
            if (__ballot(stale) != 0ull) refresh_bounds();
        }
        // screen of tile k-1 (pieces k-2, k-1, k)
        float m0[MT][2][2];
        {
            const unsigned a_at = lds_addr(abuf) + static_cast<unsigned>(k & 1) * kCoefLds + lane16;
            uint4v af[3];
#pragma unroll
            for (int k3 = 0; k3 < 3; k3++) af[k3] = lds_ld128(a_at, k3 * 1024);
            lds_wait(af[0], af[1], af[2]);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                uint4v an[3] = {af[0], af[1], af[2]};
                if (mt + 1 < MT) {
#pragma unroll
                    for (int k3 = 0; k3 < 3; k3++) an[k3] = lds_ld128(a_at, ((mt + 1) * 3 + k3) * 1024);
                }
#pragma unroll
                for (int ch = 0; ch < 2; ch++) {
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) {
                        floatx16 acc = {0};
#pragma unroll
                        for (int k3 = 0; k3 < 3; k3++)
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(af[k3]), as_half8(hist[k3][ch][nt]), acc, 0, 0, 0);
                        float m = fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
                        if (!(SMALL && mt == MT - 1)) {
#pragma unroll
                            for (int i = 4; i < 16; i += 2) m = fmaxf(fmaxf(fabsf(acc[i]), fabsf(acc[i + 1])), m);
                        }
                        m0[mt][ch][nt] = m;
                    }
                    alb[ch] = fmaxf(fmaxf(m0[mt][ch][0], m0[mt][ch][1]), alb[ch]);
                }
                if (mt + 1 < MT) {
                    lds_wait(an[0], an[1], an[2]);
#pragma unroll
                    for (int k3 = 0; k3 < 3; k3++) af[k3] = an[k3];
                }
            }
        }
        if (!passthru0 && (lb[0] == 0.0f || lb[1] == 0.0f)) refresh_bounds();  // (see k_fused_f32s)
        float margin = -__builtin_inff();
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                float mm = m0[0][ch][nt];
#pragma unroll
                for (int mt = 1; mt < MT; mt++) mm = fmaxf(mm, m0[mt][ch][nt]);
                margin = fmaxf(margin, mm - thr_l[ch][nt]);
            }
        const unsigned long long any = __ballot(margin >= 0.0f);
        if (any != 0ull) {  // wave-uniform, rare
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int ch = 0; ch < 2; ch++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) {
                        const bool flag = m0[mt][ch][nt] >= thr_l[ch][nt];
                        const unsigned long long mask = __ballot(flag);
                        if (mask == 0ull) continue;
                        const unsigned slot = n_cand + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                                                                  __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0));
                        if (flag && slot < kCandCap)
                            my_cands[slot] = make_uint2(static_cast<unsigned>(ch) | static_cast<unsigned>(32 * nt + (lane & 31)) << 1 |
                                                            static_cast<unsigned>(mt) << 7 | static_cast<unsigned>(lane >> 5) << 9 |
                                                            static_cast<unsigned>(k - 1) << 10,
                                                        __float_as_uint(m0[mt][ch][nt]));
                        n_cand += static_cast<unsigned>(__popcll(mask));
                    }
        }
    };
    for (int k = 0; k < T + 1; k += 3) {
        tp_step(k, std::integral_constant<int, 0>{});
        if (k + 1 < T + 1) tp_step(k + 1, std::integral_constant<int, 1>{});
        if (k + 2 < T + 1) tp_step(k + 2, std::integral_constant<int, 2>{});
    }
    {
        float Sf[2], Af[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            Sf[c] = wave_max_f32(s_lane[c]);
            Af[c] = wave_max_f32(alb[c]);
            lb[c] = fmaxf(lb[c], fmaxf(passthru0 ? Sf[c] : 0.0f, Af[c] - fmaf(kScreenRel, Sf[c], kScreenAbs)));
        }
        if (lane == 0) {
            cand_cnt[item_idx] = n_cand < kCandCap ? n_cand : kCandCap;
            if (n_cand > kCandCap || Sf[0] >= kF16Safe || Sf[1] >= kF16Safe) overflow[it.track] = 1u;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                atomicMax(&tpb[(static_cast<size_t>(it.track) * kMaxCh + c) * 2 + 0], __float_as_uint(lb[c]));
                atomicMax(&tpb[(static_cast<size_t>(it.track) * kMaxCh + c) * 2 + 1], __float_as_uint(Sf[c]));
                if (passthru0 && Sf[c] > 0.0f)
                    atomicMax(&peaks[static_cast<size_t>(it.track) * kMaxCh + c],
                              static_cast<unsigned long long>(__double_as_longlong(static_cast<double>(Sf[c]))));
            }
        }
    }
}

#endif  // HR_EXPERIMENTS (k_fused_split)

// Exact evaluation of the candidates that can still be the maximum.  One wave per candidate list of the fused kernel (four
// lists per workgroup), its lanes striding over the list (a few dozen entries as a rule).  A stereo item has one list; an
// item of a 5.1 / 7.1 track read directly has wpi = channels / 2, list w of item i being channels 2 w, 2 w + 1.
__global__ void __launch_bounds__(256) k_refine_cands(const TrackDesc *__restrict__ tracks,
                                                      const RateDev *__restrict__ rates,
                                                      const FusedItem *__restrict__ items, uint32_t n_lists, uint32_t wpi,
                                                      const uint2 *__restrict__ cands,
                                                      const unsigned *__restrict__ cand_cnt,
                                                      const unsigned *__restrict__ tpb,
                                                      unsigned long long *__restrict__ peaks)
{
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);  // (the list's index)
    if (item >= n_lists) return;
    if (cand_cnt[item] & kCandOverflowBit) return;  // the list ran out of slots: k_true_peak_rows redoes the whole wave
    const unsigned n = cand_cnt[item];
    const int lane = threadIdx.x & 63;
    if (n == 0) return;  // (wave-uniform)
    const FusedItem it = items[item / wpi];
    const int chan0 = 2 * static_cast<int>(item % wpi);
    const TrackDesc &t = tracks[it.track];
    const RateDev &r = rates[t.rate_idx];
    const int P = r.swr_phases, step = r.swr_step;
    // the track's proven lower bounds and the error bounds that held when a group was screened, per channel of the pair
    float lbf[2], err[2];
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        lbf[ch] = __uint_as_float(tpb[(static_cast<size_t>(it.track) * kMaxCh + chan0 + ch) * 2 + 0]);
        err[ch] = fmaf(kScreenRel, __uint_as_float(tpb[(static_cast<size_t>(it.track) * kMaxCh + chan0 + ch) * 2 + 1]), kScreenAbs);
    }
    // A listed group is 16 outputs.  Most entries cannot reach the track's bound any more (they were listed while the wave's
    // own bound was still rising) and are dropped on sight; the survivors -- a dozen per track as a rule -- are evaluated four
    // at a time, sixteen lanes each, one output per lane: a launch of many short tracks has as many survivors per track as
    // a launch of few long ones, and a lane that walked its group's sixteen outputs alone made 10 000 ten-second clips spend
    // 0.5 ms here.  (Maxima: the order of evaluation does not matter.)
    double best0 = 0.0, best1 = 0.0;
    for (unsigned base = 0; base < n; base += 64) {
        const unsigned ci = base + static_cast<unsigned>(lane);
        uint2 e = make_uint2(0u, 0u);
        bool live = false;
        if (ci < n) {
            e = cands[static_cast<size_t>(item) * kCandCap + ci];
            const int ch = e.x & 1u;
            live = !(__uint_as_float(e.y) + (ch ? err[1] : err[0]) < (ch ? lbf[1] : lbf[0]));  // can still reach the track's proven lower bound
        }
        unsigned long long mask = __ballot(live);
        while (mask != 0ull) {  // (wave-uniform)
            int src = -1;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (mask == 0ull) break;
                const int sl = __ffsll(static_cast<long long>(mask)) - 1;
                mask &= mask - 1ull;
                if ((lane >> 4) == q) src = sl;
            }
            const unsigned ex = static_cast<unsigned>(__shfl(static_cast<int>(e.x), src < 0 ? 0 : src, 64));
            if (src < 0) continue;
            const int ch = ex & 1u, crow = (ex >> 1) & 63u, mt = (ex >> 7) & 3u, half = (ex >> 9) & 1u;
            const int tile = static_cast<int>(ex >> 10);
            // (a row with a shifted piece grid, FusedItem::phi: its tile t starts phi frames before frame 16 t of the row; counted
            // from the hop in front of the row, which starts at phase 0 of the bank like every hop, nothing is negative)
            const long shift = it.phi ? static_cast<long>(it.row_len) - it.phi : 0;
            const long num0 = (16L * tile + shift) * P, num1 = (16L * (tile + 1) + shift) * P;
            const int m_lo = static_cast<int>((num0 + step - 1) / step);
            const int count = static_cast<int>((num1 + step - 1) / step) - m_lo;
            const int64_t row_frame0 = static_cast<int64_t>(it.frame0) + static_cast<int64_t>(crow) * it.row_stride * it.row_len - (it.phi ? it.row_len : 0);
            const int i = lane & 15;
            const int rowidx = 32 * mt + 4 * half + (i & 3) + 8 * (i >> 2);
            if (rowidx >= count) continue;
            const int m = m_lo + rowidx;
            if (r.tp_passthru0 && (m * step) % P == 0) continue;  // pass-through phase: taken exactly from the sample peak
            const double a = exact_output(t, r.swr_taps, P, step, row_frame0, chan0 + ch, m);
            if (ch) best1 = a > best1 ? a : best1;
            else best0 = a > best0 ? a : best0;
        }
    }
    best0 = wave_max_f64(best0);
    best1 = wave_max_f64(best1);
    if (lane == 0 && best0 > 0.0) atomicMax(&peaks[static_cast<size_t>(it.track) * kMaxCh + chan0 + 0], static_cast<unsigned long long>(__double_as_longlong(best0)));
    if (lane == 0 && best1 > 0.0) atomicMax(&peaks[static_cast<size_t>(it.track) * kMaxCh + chan0 + 1], static_cast<unsigned long long>(__double_as_longlong(best1)));
}

// ---- k_true_peak_rows: the exact true peak of whole ranges of a stereo track, fast --------------------------------------
// What redoes the positions of a fused wave whose candidate list ran out of slots (sustained tones, test signals: every
// crest is a candidate of the f16 screen), or of a whole track whose samples leave the f16 range.  The generic kernel
// (one thread per position, every thread its own row of the tap table from L2) takes 2.5 ms per 3-minute track; here
// the 32 taps of an output are wave-uniform and arrive through scalar loads as FMA operands, and the 32-frame window
// of a channel slides through registers, so an output costs its 32 f64 FMAs and little else.
//   * A lane owns redo_lane_frames(step) consecutive positions -- whole resampler cycles, so that all 64 lanes are at the
//     same phase of the bank at every step -- and a wave 64 such runs (item.first, a multiple of `step` frames; item.pad
//     positions count from there), one wave per channel.  The window of position p (frames p-15 .. p+16) sits in 32
//     registers, frame f in register (f - p0 + 15) mod 32: the loop is unrolled 32 positions deep, so every index is static.
//   * PCM arrives 8 frames per lane and load (f32 / s16, range inside the track: FAST), one load ahead; at the
//     track's edges and for packed s24 (FAST = false) frame by frame through fetch_f32s (reflection, halos, zeros).
//   * Same FMA chain as exact_output / k_true_peak_swr / the oracle (taps 0..31 in order, from 0.0): same bits.
// Replaces libswresample's swr_convert + f_ebur128's fabs max for those positions (reached from src/analyzer.rs:253-267).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(4)));

// frames f0 .. f0+7 of channel c
template <int INF, bool FAST>
__device__ __forceinline__ void tp_rows_chunk(const TrackDesc &t, int c, int64_t f0, float (&dst)[8])
{
    if constexpr (FAST && INF == 0) {
        const f4u *p = reinterpret_cast<const f4u *>(static_cast<const char *>(t.pcm) + f0 * 8);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f4u v = p[q];
            dst[2 * q] = c ? v[1] : v[0];
            dst[2 * q + 1] = c ? v[3] : v[2];
        }
    } else if constexpr (FAST && INF == 1) {
        constexpr float sc = 1.0f / 32768.0f;
        const u4u *p = reinterpret_cast<const u4u *>(static_cast<const char *>(t.pcm) + f0 * 4);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const u4u v = p[q];
#pragma unroll
            for (int e = 0; e < 4; e++)  // (one dword per frame: left in the low half)
                dst[4 * q + e] = static_cast<float>(c ? static_cast<int>(v[e]) >> 16 : static_cast<int>(v[e] << 16) >> 16) * sc;
        }
    } else {
        // (every sample of these formats, and of a neighbour shard's halo, is an f32 value: nothing is rounded here)
#pragma unroll
        for (int e = 0; e < 8; e++) dst[e] = static_cast<float>(fetch_f32s(t, c, f0 + e));
    }
}

template <int INF, bool FAST>
__device__ __forceinline__ double tp_rows_body(const TrackDesc &t, int c, cdouble_p taps, int P, int step, int K, int64_t p0,
                                               int lane_lim)
{
    double w[32], m = 0.0;
    float ca[8], cb[8];
    // the window of position p0: frames p0-15 .. p0+15 (slot j holds frame p0 - 15 + j; slot 31 is filled by position 0)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        tp_rows_chunk<INF, FAST>(t, c, p0 - 16 + 8 * q, ca);
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (8 * q + e >= 1) w[8 * q + e - 1] = static_cast<double>(ca[e]);
    }
    tp_rows_chunk<INF, FAST>(t, c, p0 + 16, ca);  // frames p0+16 .. p0+23: what positions 0 .. 7 bring into the window
    int d = 0;                                    // phase units of the next output beyond the current position (wave-uniform)
    for (int b = 0; b < K; b += 32) {
        // position b + u; false once the lane's run is over (wave-uniform).  u is a compile-time constant, so every
        // window index below is one too.
        auto position = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const int i = b + u;
            if (i >= K) return false;
            // (one chunk of 8 frames ahead of its use: A serves u mod 16 = 0..7, B serves 8..15)
            if constexpr ((u & 15) == 0) tp_rows_chunk<INF, FAST>(t, c, p0 + i + 24, cb);
            if constexpr ((u & 15) == 8) tp_rows_chunk<INF, FAST>(t, c, p0 + i + 24, ca);
            w[(u + 31) & 31] = static_cast<double>((u & 8) ? cb[u & 7] : ca[u & 7]);  // frame p0 + i + 16
            double mp = 0.0;
            while (d < P) {  // the outputs that sit at this position: phase d of the bank
                const cdouble_p f = taps + static_cast<size_t>(d) * kSwrTaps;
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < kSwrTaps; j++) v = fma(w[(u + j) & 31], f[j], v);
                asm volatile("" : "+v"(v));  // (every lane computes it: positions past a lane's limit are masked below, not branched around)
                const double a = fabs(v);
                mp = a > mp ? a : mp;
                d += step;
            }
            d -= P;
            mp = i < lane_lim ? mp : 0.0;
            m = mp > m ? mp : m;
            return true;
        };
#define HR_POS(u) if (!position(std::integral_constant<int, u>{})) break;
#define HR_POS8(u) HR_POS(u) HR_POS(u + 1) HR_POS(u + 2) HR_POS(u + 3) HR_POS(u + 4) HR_POS(u + 5) HR_POS(u + 6) HR_POS(u + 7)
        HR_POS8(0) HR_POS8(8) HR_POS8(16) HR_POS8(24)
#undef HR_POS8
#undef HR_POS
    }
    return m;
}

// one wave per (item, channel): blockIdx.x = 2 * item + channel; several waves per SIMD hide the scalar tap loads and the
// FMA chain's latency
template <int INF>
__global__ void __launch_bounds__(64) k_true_peak_rows(const TrackDesc *__restrict__ tracks, const RateDev *__restrict__ rates,
                                                           const WorkItem *__restrict__ items, unsigned long long *__restrict__ peaks)
{
    // (item.pad: positions in the low half; in the high half the first channel of the pair -- 0 for a stereo track)
    const WorkItem it = items[blockIdx.x >> 1];
    const int c = static_cast<int>(it.pad >> 16) + (blockIdx.x & 1);
    const TrackDesc &t = tracks[it.track];
    const RateDev &r = rates[t.rate_idx];
    if (!t.tp_enabled) return;
    const int lane = threadIdx.x;
    const int P = r.swr_phases, step = r.swr_step;
    const int K = static_cast<int>(redo_lane_frames(static_cast<uint32_t>(step)));
    const int64_t N = static_cast<int64_t>(t.frames);
    const int64_t first = static_cast<int64_t>(it.first);
    // positions with an output: up to the item's end, the track's end, and -- the resampler is not flushed -- tp_drop short of it
    int64_t end = first + (it.pad & 0xffffu);
    const int64_t stop = (t.stream_end ? N - t.tp_drop : N);
    if (end > stop) end = stop;
    const int64_t p0 = first + static_cast<int64_t>(lane) * K;
    const int64_t lim = end - p0;
    const int lane_lim = lim < 0 ? 0 : lim > K ? K : static_cast<int>(lim);
    const cdouble_p taps = (cdouble_p)r.swr_taps;
    // every frame any lane may fetch (16 in front of the first position, up to 80 behind the last) lies inside the track?
    const bool fast = INF != 2 && t.ch == 2 && first >= 16 && first + 64ll * K + 80 <= N;  // (the wide loads are the stereo layout's)
    double m = fast ? tp_rows_body<INF, true>(t, c, taps, P, step, K, p0, lane_lim)
                    : tp_rows_body<INF, false>(t, c, taps, P, step, K, p0, lane_lim);
    m = wave_max_f64(m);
    if (lane == 0 && m > 0.0)
        atomicMax(&peaks[static_cast<size_t>(it.track) * kMaxCh + c], static_cast<unsigned long long>(__double_as_longlong(m)));
}

void launch_true_peak_rows(int fmt, const TrackDesc *tracks, const RateDev *rates, const WorkItem *items, uint32_t n_items,
                           unsigned long long *peaks, hipStream_t st)
{
    if (!n_items) return;
    const dim3 grid(2 * n_items);  // (item, channel)
    if (fmt == kFmtS16) k_true_peak_rows<1><<<grid, dim3(64), 0, st>>>(tracks, rates, items, peaks);
    else if (fmt == kFmtS24) k_true_peak_rows<2><<<grid, dim3(64), 0, st>>>(tracks, rates, items, peaks);
    else k_true_peak_rows<0><<<grid, dim3(64), 0, st>>>(tracks, rates, items, peaks);
}

size_t fused_cand_bytes(uint32_t n_items) { return static_cast<size_t>(n_items) * kCandCap * sizeof(uint2); }

void launch_refine_cands(const TrackDesc *tracks, const RateDev *rates, const FusedItem *items, uint32_t n_lists, uint32_t wpi,
                         const FusedScratch &fs, unsigned long long *peaks, hipStream_t st)
{
    if (!n_lists) return;
    k_refine_cands<<<dim3((n_lists + 3) / 4), dim3(256), 0, st>>>(tracks, rates, items, n_lists, wpi, static_cast<const uint2 *>(fs.cands),
                                                                    fs.cand_cnt, fs.tpb, peaks);
}

template <bool DO_KW, bool DO_TP, int INF = 0>
static void launch_class(int tp_class, dim3 grid, dim3 block, size_t lds, hipStream_t st, const TrackDesc *tracks,
                         const RateDev *rates, const FusedItem *items, uint32_t n_items, HopRec *recs, uint2 *cands,
                         const FusedScratch &fs, unsigned long long *peaks, int dbg)
{
#define HR_FUSED_CASE(MT, SMALL)                                                                                      \
    k_fused_f32s<DO_KW, DO_TP, MT, SMALL, INF><<<grid, block, lds, st>>>(tracks, rates, items, n_items, recs, cands, \
                                                                          fs.cand_cnt, fs.tpb, fs.overflow, peaks, dbg)
#ifdef HR_FAST_BUILD
    if constexpr (DO_KW && DO_TP && INF == 0) { HR_FUSED_CASE(HR_FAST_MT, HR_FAST_SMALL); }
    (void)tp_class;
#else
    if constexpr (!DO_TP) {
        HR_FUSED_CASE(1, false);
    } else {
        switch (tp_class) {
        case 0: HR_FUSED_CASE(1, false); break;
        case 1: HR_FUSED_CASE(1, true); break;
        case 2: HR_FUSED_CASE(2, false); break;
        case 3: HR_FUSED_CASE(2, true); break;
        case 4: HR_FUSED_CASE(3, false); break;
        default: HR_FUSED_CASE(3, true); break;
        }
    }
#endif
#undef HR_FUSED_CASE
}

int fused_tp_class(int tp_mt, int tp_last_rows) { return (tp_mt - 1) * 2 + (tp_last_rows <= 8 ? 1 : 0); }

#ifdef HR_EXPERIMENTS
template <int MT, bool SMALL, int PAIRS, int RING>
static void launch_split_geom(uint32_t n_items, hipStream_t st, const TrackDesc *tracks, const RateDev *rates,
                              const FusedItem *items, HopRec *recs, uint2 *cands, const FusedScratch &fs,
                              unsigned long long *peaks, int wave_map)
{
    constexpr int lds = split_lds(PAIRS, RING);
    // more than 64 KiB of dynamic LDS has to be asked for (per device: contexts of one process may sit on several GPUs)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused_split<MT, SMALL, PAIRS, RING>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k_fused_split<MT, SMALL, PAIRS, RING><<<dim3(n_items / PAIRS), dim3(PAIRS * 128), lds, st>>>(
        tracks, rates, items, recs, cands, fs.cand_cnt, fs.tpb, fs.overflow, peaks, wave_map);
}

// HR_SPLIT_GEOM: 0 = four pairs per workgroup, ring of four piece buffers (148 KiB LDS: one workgroup per CU);
// 1 = two pairs, ring of two (52 KiB, <= 168 VGPRs: three workgroups per CU = three waves per SIMD)
template <int MT, bool SMALL>
static void launch_split_case(dim3 grid, hipStream_t st, const TrackDesc *tracks, const RateDev *rates, const FusedItem *items,
                              HopRec *recs, uint2 *cands, const FusedScratch &fs, unsigned long long *peaks)
{
    static const int wave_map = std::getenv("HR_WAVE_MAP") ? std::atoi(std::getenv("HR_WAVE_MAP")) : 0;
    static const int geom = std::getenv("HR_SPLIT_GEOM") ? std::atoi(std::getenv("HR_SPLIT_GEOM")) : 0;
    const uint32_t n_items = grid.x * 4;  // (the host pads every run of items to a multiple of four)
    if (geom == 1) launch_split_geom<MT, SMALL, 2, 2>(n_items, st, tracks, rates, items, recs, cands, fs, peaks, wave_map);
    else launch_split_geom<MT, SMALL, 4, 4>(n_items, st, tracks, rates, items, recs, cands, fs, peaks, wave_map);
}

#endif  // HR_EXPERIMENTS (split-wave launchers)

void launch_fused_f32s(const TrackDesc *tracks, const RateDev *rates, const FusedItem *items, uint32_t n_items,
                       HopRec *recs, const FusedScratch &fs, unsigned long long *peaks, int variant, int tp_class, int dbg,
                       hipStream_t st)
{
    if (!n_items) return;
    const dim3 grid((n_items + kFusedWaves - 1) / kFusedWaves), block(64 * kFusedWaves);
    // (the data-movement-only ablation build takes just its piece buffers: more workgroups per CU, which shows how
    // the stream rate depends on the bytes in flight)
    size_t lds = variant == 3 ? kFusedWaves * kPieceLds : kGroupLds;
#ifdef HR_EXPERIMENTS
    static const int lds_pad = std::getenv("HR_EXP_LDS_PAD") ? std::atoi(std::getenv("HR_EXP_LDS_PAD")) : 0;  // occupancy experiments
    lds += lds_pad;
    if (lds > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused_f32s<true, true, HR_FAST_MT_OR_3, HR_FAST_SMALL_OR_TRUE, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
#endif
    uint2 *cands = static_cast<uint2 *>(fs.cands);
#if defined(HR_EXPERIMENTS) && !defined(HR_FAST_BUILD)
    if (variant == 4) {  // full pass, K-weighting and true peak in separate waves
        switch (tp_class) {
        case 0: launch_split_case<1, false>(grid, st, tracks, rates, items, recs, cands, fs, peaks); break;
        case 1: launch_split_case<1, true>(grid, st, tracks, rates, items, recs, cands, fs, peaks); break;
        case 2: launch_split_case<2, false>(grid, st, tracks, rates, items, recs, cands, fs, peaks); break;
        case 3: launch_split_case<2, true>(grid, st, tracks, rates, items, recs, cands, fs, peaks); break;
        case 4: launch_split_case<3, false>(grid, st, tracks, rates, items, recs, cands, fs, peaks); break;
        default: launch_split_case<3, true>(grid, st, tracks, rates, items, recs, cands, fs, peaks); break;
        }
        launch_refine_cands(tracks, rates, items, n_items, 1, fs, peaks, st);
        return;
    }
#endif
    if (variant == 5 || variant == 6) {  // full pass on stereo s16 / s24
        if (variant == 5) launch_class<true, true, 1>(tp_class, grid, block, lds, st, tracks, rates, items, n_items, recs, cands, fs, peaks, dbg);
        else launch_class<true, true, 2>(tp_class, grid, block, lds, st, tracks, rates, items, n_items, recs, cands, fs, peaks, dbg);
        launch_refine_cands(tracks, rates, items, n_items, 1, fs, peaks, st);
        return;
    }
    switch (variant) {
    case 1: launch_class<true, false>(0, grid, block, lds, st, tracks, rates, items, n_items, recs, cands, fs, peaks, dbg); break;
    case 2: launch_class<false, true>(tp_class, grid, block, lds, st, tracks, rates, items, n_items, recs, cands, fs, peaks, dbg); break;
    case 3: launch_class<false, false>(0, grid, block, lds, st, tracks, rates, items, n_items, recs, cands, fs, peaks, dbg); break;
    default: launch_class<true, true>(tp_class, grid, block, lds, st, tracks, rates, items, n_items, recs, cands, fs, peaks, dbg); break;
    }
    if (variant == 0 || variant == 2)
        launch_refine_cands(tracks, rates, items, n_items, 1, fs, peaks, st);
}

}  // namespace hr
