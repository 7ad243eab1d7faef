// Device-resident batch prover / verifier of the engine — product code (see engine.hpp, protocol.h).
//
// For batches of >= 56 proofs (option device_min_batch) the whole protocol runs on the GPU: the MSM phases read their
// scalars from device memory through task descriptors that are built ONCE per loaded batch shape ("plans"), the
// finalisation kernels leave the compressed results in a per-proof slot registry, and one-wave-per-proof step kernels
// (protocol.hip) hash them into the transcripts and derive the next phase's scalars.  The host only enqueues: no
// synchronisation between cpx_batch_prove's input upload and the download of the proofs.  (Small batches keep the
// host-driven path of engine.cpp: a lone transcript is latency-bound on a GPU wave.)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "engine.hpp"

namespace cpx {

// ---------------------------------------------------------------- plans
// Layout of a table-backed MSM phase whose requests all read device-resident scalars (TblReq::dev): the CRS segments go to
// k_msm_fix, the per-proof segments to k_msm_tblw; k_reduce_sets turns the raw lane accumulators into partial sums,
// k_finalize_ranges adds them per request, normalises, scatters the affine point (TblReq::dst) and writes the compressed
// bytes to slot `comp_index[i]` of the registry.
void Engine::build_plan(TblPlan& pl, const std::vector<TblReq>& reqs, const std::vector<uint32_t>& comp_index) {
  const size_t nt = reqs.size();
  const bool fix = fix_bits_ && fixtab();
  if (!fix) throw std::logic_error("set_crs first");
  const TAff* c_lo = ctab();
  const TAff* c_hi = ctab() + (size_t)copies_ * nc();
  auto is_crs = [&](const TblSeg& sg) { return sg.n && sg.base >= c_lo && sg.base < c_hi; };
  auto needs_tbl = [&](const TblReq& r) { return (r.seg0.n && !is_crs(r.seg0)) || (r.seg1.n && !is_crs(r.seg1)); };
  size_t ntt = 0, nft = 0;
  for (const TblReq& r : reqs) {
    if (!r.dev && (r.seg0.n || r.seg1.n)) throw std::logic_error("device plan: request without device scalars");
    ntt += needs_tbl(r) ? 1 : 0;
    nft += (is_crs(r.seg0) ? 1 : 0) + (is_crs(r.seg1) ? 1 : 0);
  }
  pl.nt = nt;
  pl.ntt = ntt;
  pl.nft = nft;
  pl.fix_wpw = pl.force_fix_wpw ? pl.force_fix_wpw : msm_fix_windows_per_wave(opt_, (int)nft, fix_bits_);
  pl.tbl_wpw = pl.force_tbl_wpw ? pl.force_tbl_wpw : msm_tblw_windows_per_wave(opt_, (int)ntt);
  const uint32_t fix_parts = (uint32_t)msm_fix_parts(fix_bits_, pl.fix_wpw), tbl_parts = (uint32_t)msm_tblw_parts(pl.tbl_wpw);
  std::vector<TblTask> ht(ntt);
  std::vector<FixTask> hf(nft);
  std::vector<uint32_t> meta(7 * nt);   // pfirst | pcount | dst | comp_index | addends[3]
  // requests without a destination scatter their affine point to a write-only slot of proof 0 that NOTHING ever reads: TMP(7) for plans of the
  // main stream, TMP(5) for the plan that runs beside them on the table stream (phase 1t) — two streams never write the same slot
  const uint32_t dummy_dst = slot_index(0, SlotMap(L_).TMP(pl.table_stream ? 5 : 7));
  const TblSeg none{nullptr, nullptr, 0, 0};
  size_t it = 0, jf = 0, nparts = 0;
  pl.pts_fix = pl.pts_tbl = 0;
  pl.any_add = false;
  for (size_t i = 0; i < nt; i++) {
    const TblReq& r = reqs[i];
    const bool f0 = is_crs(r.seg0), f1 = is_crs(r.seg1);
    const uint32_t first = (uint32_t)nparts;
    if (needs_tbl(r)) {
      TblTask t;
      t.seg[0] = r.seg0;
      t.seg[1] = f1 ? none : r.seg1;
      t.scalars = r.dev;
      if (f0) {   // seg0 goes to the fixed-base kernel: its scalars are skipped, seg1 becomes the only segment
        t.seg[0] = f1 ? none : r.seg1;
        t.seg[1] = none;
        t.scalars = r.dev + r.seg0.n;
      }
      t.flags = 0;
      t.pad = (uint32_t)nparts;
      t.digits = nullptr;
      ht[it++] = t;
      nparts += tbl_parts;
    }
    if (f0) {
      hf[jf++] = FixTask{r.seg0.idx, r.dev, (uint32_t)(r.seg0.base - c_lo), r.seg0.n, 0, (uint32_t)nparts};
      nparts += fix_parts;
    }
    if (f1) {
      hf[jf++] = FixTask{r.seg1.idx, r.dev + r.seg0.n, (uint32_t)(r.seg1.base - c_lo), r.seg1.n, 0, (uint32_t)nparts};
      nparts += fix_parts;
    }
    meta[i] = first;
    meta[nt + i] = (uint32_t)nparts - first;
    meta[2 * nt + i] = r.dst != ~0u ? r.dst : dummy_dst;
    meta[3 * nt + i] = comp_index[i];
    for (int j = 0; j < 3; j++) meta[4 * nt + 3 * i + j] = r.add[j];
    pl.any_add |= r.add[0] != ~0u;
    pl.pts_fix += (f0 ? r.seg0.n : 0) + (f1 ? r.seg1.n : 0);
    pl.pts_tbl += (f0 ? 0 : r.seg0.n) + (f1 ? 0 : r.seg1.n);
  }
  // Long tasks first: a work-group's task is its index in these arrays, partial-sum slots travel with the task (out_first / pad).  With the
  // requests in protocol order a launch ended with the long waves of the last proofs and the GPU drained behind them (~0.8 ms of a
  // 7.6-ms k_msm_fix launch at 8192 proofs); now the one-point tasks fill the tail.
  if (!pl.keep_order) {
    std::stable_sort(hf.begin(), hf.end(), [](const FixTask& a, const FixTask& b) { return a.n > b.n; });
    std::stable_sort(ht.begin(), ht.end(), [](const TblTask& a, const TblTask& b) { return a.seg[0].n + a.seg[1].n > b.seg[0].n + b.seg[1].n; });
  }
  pl.nparts = nparts;
  pl.fix_sets = nft * fix_parts;
  pl.tbl_sets = ntt * tbl_parts;
  pl.ttasks.ensure(std::max<size_t>(ntt, 1));
  pl.ftasks.ensure(std::max<size_t>(nft, 1));
  pl.meta.ensure(std::max<size_t>(meta.size(), 1));
  if (ntt) CPX_HIP(hipMemcpy(pl.ttasks.p, ht.data(), ntt * sizeof(TblTask), hipMemcpyHostToDevice));
  if (nft) CPX_HIP(hipMemcpy(pl.ftasks.p, hf.data(), nft * sizeof(FixTask), hipMemcpyHostToDevice));
  if (nt) CPX_HIP(hipMemcpy(pl.meta.p, meta.data(), meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
}

void Engine::exec_plan(const TblPlan& pl, uint8_t* d_comp_registry, bool on_table_stream) {
  if (!pl.nt) return;
  // (a plan on the table stream works in the table stream's own scratch, sized by prepare_device_prover: it runs beside the plans
  // of the main stream)
  const bool tl = on_table_stream && table_stream_on();
  hipStream_t const st = tl ? tab_.dstream : stream_;
  if (!tl) {
    d_tpart_.ensure(std::max<size_t>(pl.nparts, 1));
    ensure_raw_sets(pl.fix_sets + pl.tbl_sets);
  }
  uint32_t* const raw = tl ? tab_.raw.p : d_raw_.p;
  uint32_t* const rawslot = tl ? tab_.rawslot.p : d_rawslot_.p;
  TJac* const part = tl ? tab_.part.p : d_tpart_.p;
  if (pl.nft) {
    tick(fix_kernel_name(fix_bits_, pl.fix_wpw), 128.0 * pl.pts_fix, pl.pts_fix);
    launch_msm_fix(pl.ftasks.p, (int)pl.nft, fixtab(), fix_bits_, pl.fix_wpw, (int)nc(), raw, rawslot, st);
    tock();
  }
  if (pl.ntt) {
    tick(tblw_kernel_name(pl.tbl_wpw), 128.0 * pl.pts_tbl, pl.pts_tbl);
    launch_msm_tblw(pl.ttasks.p, (int)pl.ntt, pl.tbl_wpw, raw + pl.fix_sets * raw_set_words(), rawslot + pl.fix_sets, st);
    tock();
  }
  if (tl) launch_reduce_sets(opt_, raw, rawslot, (int)pl.fix_sets, (int)pl.tbl_sets, tab_.mid.p, part, st, (int)B_);   // (not timed: a span is bracketed on the main stream)
  else reduce_sets(pl.fix_sets, pl.tbl_sets);
  const uint32_t* m = pl.meta.p;
  tick("k_finalize_ranges", 0, (double)pl.nt);
  launch_finalize_ranges(opt_, part, m, m + pl.nt, (int)pl.nt, d_pp_.p, m + 2 * pl.nt, d_comp_registry, st, pl.any_add ? m + 4 * pl.nt : nullptr, m + 3 * pl.nt);
  tock();
}

// One late round: lane-per-output Straus MSMs over the materialised points, then the usual finalisation (affine, compressed bytes
// into the slot registry; nothing reads these points in affine form again)
void Engine::exec_late_round(const LateRound& r0, size_t comp_off, const char* what) {
  LateRound r = r0;
  const DevProver::Late& lt = dprove_.late;
  r.out = lt.part.p;
  const int nreq = r.total / r.slices, hm = r.m / 2, Bi = r.nproofs;
  const double pairs = (double)nreq * hm;
  tick(what, 128.0 * pairs, pairs);
  launch_late_msm(r, stream_);
  tock();
  const bool ipa = r.nout == 4;
  if (ipa) {   // beta <c_L, d_R> H and beta <c_R, d_L> H (inner_product_argument.rs:152,158) from the table of multiples: the extras of L_C, R_C
    launch_late_fix(d_rout_.p + hm, r.scal_proof_stride, lt.h_col, 1, 1, fixtab(), fix_bits_, (int)nc(), lt.extra.p, 4, Bi, stream_);
    launch_late_fix(d_rout_.p + 3 * hm + 1, r.scal_proof_stride, lt.h_col, 1, 1, fixtab(), fix_bits_, (int)nc(), lt.extra.p + 2, 4, Bi, stream_);
  }
  tick("k_msm_tail", 0, (double)nreq);
  launch_msm_tail(opt_, lt.part.p, d_tpart_.p, nullptr, nreq, r.slices, 128 / r.slices, stream_, ipa ? lt.extra.p : nullptr, ipa ? 1 : 0);
  tock();
  const uint32_t* meta = lt.meta.p;
  tick("k_finalize_ranges", 0, (double)nreq);
  launch_finalize_ranges(opt_, d_tpart_.p, meta, meta + 6 * B_, nreq, nullptr, nullptr, dprove_.slotcomp.p, stream_, nullptr, meta + comp_off);
  tock();
}

// ---------------------------------------------------------------- prover
// (Re)builds everything that depends only on the shape of the loaded batch: device buffers, the ProveDev view, the task
// descriptors of every phase and of the side stream.
void Engine::prepare_device_prover() {
  const size_t B = B_, ell = ell_, n = n_, L = L_;
  const SlotMap sm(L);
  const RandIdx ri((int)n);
  const size_t NS = sm.count(), NP = np(), nrand = ri.count();
  DevProver& dp = dprove_;
  // buffers (sizes only grow; a reallocation changes a pointer and invalidates the plans)
  d_bytes_.ensure(B * 4 * ell * 48);
  d_mcomp_.ensure(B * 48);
  d_tstate_.ensure(B * 27);
  d_veca_.ensure(B * ell);
  d_rvec_.ensure(B * 4 * n);
  dp.rvec2.ensure(B * 2 * n);
  d_rgam_.ensure(B * 2);
  d_rbeta_.ensure(B);
  d_rout_.ensure(B * (2 * n + 2));
  dp.perm.ensure(B * ell);
  dp.k.ensure(B);
  dp.mbl.ensure(B * 4);
  dp.rnd.ensure(B * nrand);
  dp.vec.ensure(B * (size_t)V_COUNT * n);
  dp.sc.ensure(B * (size_t)SC_COUNT);
  dp.slotcomp.ensure(B * NS * 48);
  dp.proofs.ensure(B * proof_size());
  dp.mdst.ensure(B);
  side_.tasks.ensure(2 * B);
  side_.stasks.ensure(4 * B);
  side_.res.ensure(2 * B);
  side_.dst.ensure(2 * B);
  side_.conv.ensure(4 * B * ell);
  side_.ttasks.ensure(2 * B);
  side_.digits.ensure(9 * 2 * B * ell);
  side_.raw.ensure(2 * B * 32 * raw_set_words());
  side_.rawslot.ensure(2 * B * 32);
  side_.mid.ensure(2 * B * 32 * reduce_mid_per_set());
  side_.part.ensure(2 * B * 32);
  const std::vector<const void*> sig = {d_pp_.p,     d_ptab_.p,    d_psrc_.p,  d_bytes_.p,  d_mcomp_.p,   d_tstate_.p,   d_veca_.p,      d_rvec_.p,
                                        dp.rvec2.p,  d_rgam_.p,    d_rbeta_.p, d_rout_.p,   dp.perm.p,    dp.k.p,        dp.mbl.p,       dp.rnd.p,
                                        dp.vec.p,    dp.sc.p,      dp.slotcomp.p, dp.proofs.p, side_.tasks.p, side_.stasks.p, side_.dst.p, side_.conv.p,
                                        ctab(),      fixtab(),     (const void*)(uintptr_t)B, (const void*)(uintptr_t)ell, (const void*)(uintptr_t)fix_bits_};
  if (sig == dp.signature) return;
  dp.signature.clear();

  ProveDev& d = dp.dev;
  d.ell = (int)ell;
  d.n = (int)n;
  d.L = (int)L;
  d.NS = (int)NS;
  d.psz = proof_size();
  d.perm = dp.perm.p;
  d.k = dp.k.p;
  d.mbl = dp.mbl.p;
  d.rnd = dp.rnd.p;
  d.tstate = d_tstate_.p;
  d.veca = d_veca_.p;
  d.vec = dp.vec.p;
  d.sc = dp.sc.p;
  d.slotcomp = dp.slotcomp.p;
  d.inst_comp = d_bytes_.p;
  d.mcomp = d_mcomp_.p;
  d.rvec = d_rvec_.p;
  d.rvec2 = dp.rvec2.p;
  d.rgam = d_rgam_.p;
  d.rbeta = d_rbeta_.p;
  d.proofs = dp.proofs.p;
  memcpy(d.crs_h_comp, crs_H_comp_, 48);

  auto rnd = [&](size_t p, int i) { return dp.rnd.p + p * nrand + i; };
  auto vec = [&](size_t p, int v) { return dp.vec.p + (p * V_COUNT + v) * n; };
  auto sc = [&](size_t p, int i) { return dp.sc.p + p * SC_COUNT + i; };
  auto cidx = [&](size_t p, int slot) { return (uint32_t)(p * NS + slot); };
  const size_t O_M = 0, O_T = 1, O_U = 1 + n;
  const uint32_t* iH = idx_list({(uint32_t)n});
  const uint32_t* iGt = idx_list({(uint32_t)(n + 1)});
  const uint32_t* iGu = idx_list({(uint32_t)(n + 2)});
  std::vector<uint32_t> gb(n);   // SameMSM basis G | Hvec[0..2) | G_t | G_u as CRS-table indices
  for (size_t i = 0; i < ell + 2; i++) gb[i] = (uint32_t)i;
  gb[ell + 2] = (uint32_t)(n + 1);
  gb[ell + 3] = (uint32_t)(n + 2);
  const uint32_t* iGb = idx_list(gb);
  const TblSeg none{nullptr, nullptr, 0, 0};
  auto req = [&](TblSeg s0, TblSeg s1, const Fr* dev, uint32_t dst = ~0u) {
    TblReq r{s0, nullptr, s1, nullptr, dst};
    r.dev = dev;
    return r;
  };
  std::vector<TblReq> reqs;
  std::vector<uint32_t> ci;

  // mdst: where the affine M of proof p goes (table source slot 0)
  {
    std::vector<uint32_t> md(B);
    for (size_t p = 0; p < B; p++) md[p] = (uint32_t)(p * NP + O_M);
    CPX_HIP(hipMemcpy(dp.mdst.p, md.data(), B * 4, hipMemcpyHostToDevice));
  }
  // (decided before the plans are laid out: the phases of a small batch differ, too)
  const size_t hn = n / 2, per = 4 * hn + 2;
  // the last log2(m) rounds of a large batch work on m materialised folded bases per family instead (late.hip): m = 16 (four rounds) up to
  // n = 256; an all-MSM round of n = 512 / 1024 costs two / four times as much, while materialising costs the same whatever m (every base
  // is visited once) and a late round 32 x (4 + m) operations per cross term: m = 32 (five rounds) pays there — measured per pass of 2048
  // proofs of ell = 1020: m = 16 534.8 ms, m = 32 506.1 ms, m = 64 506.7 ms (six rounds, two k_late_uniform waves per proof: no further gain);
  // 4096 proofs of ell = 508: 500.9 / 475.8 / 495.2 ms; 8192 proofs of ell = 252: 507.4 / 511.2 ms
  DevProver::Late& lt = dp.late;
  lt.m = opt_.late_m ? (int)opt_.late_m : n >= 512 ? 32 : 16;
  while (lt.m > 16 && !late_supported((int)n, lt.m)) lt.m /= 2;
  lt.nr = 0;
  while ((1 << lt.nr) < lt.m) lt.nr++;
  // (late_min_batch is stated for n <= 256; larger proofs have larger grids per proof: the threshold shrinks with 256 / n)
  const size_t late_min = n <= 256 ? (size_t)opt_.late_min_batch : std::max<size_t>(1, (size_t)opt_.late_min_batch * 256 / n);
  lt.on = opt_.late_rounds != 0 && L >= (size_t)lt.nr + 1 && B >= late_min && late_supported((int)n, lt.m);
  lt.j0 = lt.on ? L - (size_t)lt.nr : L;
  // fused log rounds (round.hip): every round of both arguments is ONE launch — for the batches in which a round is a chain of latency-bound
  // kernels (below the late rounds' threshold; the 16-bit table of multiples)
  const size_t fused_max = n <= 256 ? (size_t)opt_.fused_rounds_max : (size_t)opt_.fused_rounds_max * 256 / n;
  const size_t fused_smsm_max = n <= 256 ? (size_t)opt_.fused_smsm_max : (size_t)opt_.fused_smsm_max * 256 / n;
  const bool fused_ok = !lt.on && fix_bits_ == 16 && !opt_.serial_streams;
  dp.fused = fused_ok && opt_.fused_rounds_max > 0 && B <= fused_max;
  dp.fused_smsm = fused_ok && opt_.fused_smsm_max > 0 && B <= fused_smsm_max;
  // wave shapes of the fused rounds: as many waves per proof as find a SIMD of their own (1024 SIMDs; proofs of n > 256 take n / 256 times the work)
  const size_t simd_share = 1024 * 256 / (B * std::max<size_t>(n, 256));   // SIMDs per proof
  const int f_fix_ipa = opt_.fused_fix_wpw ? (int)opt_.fused_fix_wpw : (simd_share >= 16 ? 4 : simd_share >= 8 ? 8 : 16);   // 16 / 8 / 4 waves per proof
  // SameMSM: separate fixed-base and table waves (12 per proof) while they find SIMDs of their own; then combined waves — a fixed-base block
  // and a table block per wave — 8 per proof (4 + 16 windows), and 4 per proof (8 + 32 windows) from 129 proofs on
  const bool f_combine = opt_.fused_combine >= 0 ? opt_.fused_combine != 0 : simd_share < 12;
  const bool f_combine4 = f_combine && opt_.fused_combine < 0 && simd_share < 8;
  const int f_fix_smsm = f_combine ? (f_combine4 ? 8 : 4) : std::max(8, opt_.fused_fix_wpw ? (int)opt_.fused_fix_wpw : 8);
  const int f_tbl_smsm = f_combine ? (f_combine4 ? 32 : 16) : (int)opt_.fused_tbl_wpw;
  dp.fused = dp.fused && round_fused_supported(f_fix_ipa, 0, true);
  dp.fused_smsm = dp.fused_smsm && round_fused_supported(f_fix_smsm, f_tbl_smsm, false);
  // -- phase 1: everything that depends only on vec_a and the prover's randomness (curdleproofs.rs:93,110-116,
  //    same_multiscalar_argument.rs:80 (B_a; B_t and B_u are phase 1t), inner_product_argument.rs:126, same_scalar_argument.rs:60-61).
  for (size_t p = 0; p < B; p++) {
    const int TMPX = sm.TMP(6);   // compressed bytes nobody reads
    if (!opt_.p1_split) {
      reqs.push_back(req(cseg(0, (uint32_t)n), none, vec(p, V_APERM), slot_index(p, SL_A)));          // A
      ci.push_back(cidx(p, SL_A));
    }
    reqs.push_back(req(cseg(0, (uint32_t)n, iGb), none, rnd(p, ri.VR())));                            // B_a
    ci.push_back(cidx(p, sm.BA()));
    reqs.push_back(req(cseg(0, (uint32_t)n), none, rnd(p, ri.IR())));                                 // B_c = msm(G | Hvec, r_c)
    ci.push_back(cidx(p, SL_BC));
    reqs.push_back(req(cseg(0, 1, iGt), none, rnd(p, ri.RT()), slot_index(p, SL_CMT1)));              // cm_T.T_1 = r_t G_t
    ci.push_back(cidx(p, SL_CMT1));
    reqs.push_back(req(cseg(0, 1, iGu), none, rnd(p, ri.RU()), slot_index(p, SL_CMU1)));              // cm_U.T_1
    ci.push_back(cidx(p, SL_CMU1));
    reqs.push_back(req(cseg(0, 1, iGt), none, rnd(p, ri.RA())));                                      // cm_A.T_1
    ci.push_back(cidx(p, sm.CMA1()));
    reqs.push_back(req(cseg(0, 1, iGu), none, rnd(p, ri.RB())));                                      // cm_B.T_1
    ci.push_back(cidx(p, sm.CMB1()));
    const int rs[4] = {ri.RT(), ri.RU(), ri.RA(), ri.RB()};                                           // r * H halves of the four T_2 commitments
    for (int q = 0; q < 4; q++) {
      reqs.push_back(req(cseg(0, 1, iH), none, rnd(p, rs[q]), slot_index(p, sm.TMP(q))));
      ci.push_back(cidx(p, TMPX));
    }
  }
  build_plan(dp.p1, reqs, ci);
  // -- phase 1t: B_t = msm(T_b, r), B_u = msm(U_b, r) (same_multiscalar_argument.rs:81-82) — the two commitments of phase 1 over the
  //    per-proof tables.  Nothing reads them before the SameMSM transcript step: they follow the table build on the table stream, and
  //    the main stream goes on with the CRS-only commitments at once
  reqs.clear();
  ci.clear();
  for (size_t p = 0; p < B; p++) {
    reqs.push_back(req(pseg(p, O_T, (uint32_t)n), none, rnd(p, ri.VR())));
    ci.push_back(cidx(p, sm.BT()));
    reqs.push_back(req(pseg(p, O_U, (uint32_t)n), none, rnd(p, ri.VR())));
    ci.push_back(cidx(p, sm.BU()));
  }
  dp.p1t.table_stream = true;
  build_plan(dp.p1t, reqs, ci);
  tab_.raw.ensure(std::max<size_t>(dp.p1t.fix_sets + dp.p1t.tbl_sets, 1) * raw_set_words());
  tab_.rawslot.ensure(std::max<size_t>(dp.p1t.fix_sets + dp.p1t.tbl_sets, 1));
  tab_.mid.ensure(std::max<size_t>(dp.p1t.fix_sets + dp.p1t.tbl_sets, 1) * reduce_mid_per_set());
  tab_.part.ensure(std::max<size_t>(dp.p1t.nparts, 1));
  // -- phase 1b (option p1_split): A = msm(G | Hvec, a_sigma | blinders) (curdleproofs.rs:93) alone — the one commitment of phase 1
  //    that needs vec_a; the rest of the phase then runs before the main stream waits for the transcript prefix
  reqs.clear();
  ci.clear();
  if (opt_.p1_split)
    for (size_t p = 0; p < B; p++) {
      reqs.push_back(req(cseg(0, (uint32_t)n), none, vec(p, V_APERM), slot_index(p, SL_A)));
      ci.push_back(cidx(p, SL_A));
    }
  build_plan(dp.p1b, reqs, ci);
  // -- phase 2: B = A + alpha M + beta sum(G) (same_permutation_argument.rs:75-76), A' = A + cm_T.T_1 + cm_U.T_1
  //    (curdleproofs.rs:134), C = msm(G | Hvec, c) (grand_product_argument.rs:76)
  reqs.clear();
  ci.clear();
  for (size_t p = 0; p < B; p++) {
    // B.  Large batches: A + alpha M + beta sum(G) — two points and an addend.  Fused / small batches: the commitment the reference computes,
    // msm(G | Hvec, a_sigma + alpha sigma + beta | a_blinders + alpha m_blinders) (same_permutation_argument.rs:75-76; the scalars are the
    // grand-product factors k_ps_sameperm leaves in V_FACT) — a 256-point task of the fixed-base kernel like C beside it, instead of a
    // one-point task of the bucket-list kernel whose 32 bucket sets per proof cost the phase 0.5 ms of reductions at 128 proofs
    TblReq rb = dp.fused_smsm ? req(cseg(0, (uint32_t)n), none, vec(p, V_FACT), slot_index(p, SL_B))
                         : req(cseg(n + 3, 1), pseg(p, O_M, 1), sc(p, SC_BETA_SP), slot_index(p, SL_B));
    if (!dp.fused_smsm) rb.add[0] = slot_index(p, SL_A);
    reqs.push_back(rb);
    ci.push_back(cidx(p, SL_B));
    TblReq ra = req(none, none, nullptr);
    ra.add[0] = slot_index(p, SL_A);
    ra.add[1] = slot_index(p, SL_CMT1);
    ra.add[2] = slot_index(p, SL_CMU1);
    reqs.push_back(ra);
    ci.push_back(cidx(p, sm.APRIME()));
    reqs.push_back(req(cseg(0, (uint32_t)n), none, vec(p, V_C)));
    ci.push_back(cidx(p, SL_C));
  }
  build_plan(dp.p2, reqs, ci);
  // -- phase 3: D = B - beta^-1 sum(G) + alpha sum(H) (grand_product_argument.rs:132), B_d = msm(G', r_d) = msm(G, r_d o u)
  reqs.clear();
  ci.clear();
  for (size_t p = 0; p < B; p++) {
    TblReq rd = req(cseg(n + 3, 2), none, sc(p, SC_NEG_BETA_G_INV));
    rd.add[0] = slot_index(p, SL_B);
    reqs.push_back(rd);
    ci.push_back(cidx(p, sm.D()));
    reqs.push_back(req(cseg(0, (uint32_t)n), none, vec(p, V_ZZU)));
    ci.push_back(cidx(p, SL_BD));
  }
  build_plan(dp.p3, reqs, ci);
  // -- IPA rounds as MSMs over the original bases; scalars from k_ipa_round_scalars
  dp.ipa.clear();
  dp.smsm.clear();
  for (size_t j = 0; j < L; j++) {
    if (j >= lt.j0) {   // placeholders: exec_late_round takes these rounds
      dp.ipa.emplace_back(nullptr);
      dp.smsm.emplace_back(nullptr);
      continue;
    }
    const size_t half = n >> (j + 1);
    std::vector<uint32_t> hi, lo, ghi, glo;
    for (size_t k = 0; k < n; k++) {
      ((k & half) ? hi : lo).push_back((uint32_t)k);
      ((k & half) ? ghi : glo).push_back(gb[k]);
    }
    const uint32_t *d_hi = idx_list(hi), *d_lo = idx_list(lo), *d_ghi = idx_list(ghi), *d_glo = idx_list(glo);
    // (fused rounds: the H term of L_C / R_C rides in the same fixed-base task as the n/2 bases — one more column of the gather list, the
    // scalar beta <c, d> follows the n/2 cross-term scalars in memory anyway — instead of a one-point task of its own: 4 tasks per proof)
    const uint32_t *d_hiH = nullptr, *d_loH = nullptr;
    if (dp.fused) {
      std::vector<uint32_t> hiH(hi), loH(lo);
      hiH.push_back((uint32_t)n);
      loH.push_back((uint32_t)n);
      d_hiH = idx_list(hiH);
      d_loH = idx_list(loH);
    }
    reqs.clear();
    ci.clear();
    for (size_t p = 0; p < B; p++) {
      const Fr* o = d_rout_.p + p * per;   // [L_C (hn), beta<c_L,d_R> | L_D (hn) | R_C (hn), beta<c_R,d_L> | R_D (hn)]
      if (dp.fused) reqs.push_back(req(cseg(0, (uint32_t)hn + 1, d_hiH), none, o));
      else reqs.push_back(req(cseg(0, (uint32_t)hn, d_hi), cseg(0, 1, iH), o));
      ci.push_back(cidx(p, sm.LC((int)j)));
      reqs.push_back(req(cseg(0, (uint32_t)hn, d_lo), none, o + hn + 1));
      ci.push_back(cidx(p, sm.LD((int)j)));
      if (dp.fused) reqs.push_back(req(cseg(0, (uint32_t)hn + 1, d_loH), none, o + 2 * hn + 1));
      else reqs.push_back(req(cseg(0, (uint32_t)hn, d_lo), cseg(0, 1, iH), o + 2 * hn + 1));
      ci.push_back(cidx(p, sm.RC((int)j)));
      reqs.push_back(req(cseg(0, (uint32_t)hn, d_hi), none, o + 3 * hn + 2));
      ci.push_back(cidx(p, sm.RD((int)j)));
    }
    dp.ipa.emplace_back(new TblPlan());
    if (dp.fused) {
      dp.ipa.back()->keep_order = true;
      dp.ipa.back()->force_fix_wpw = f_fix_ipa;
    }
    build_plan(*dp.ipa.back(), reqs, ci);
    reqs.clear();
    ci.clear();
    for (size_t p = 0; p < B; p++) {
      const Fr* oL = d_rout_.p + p * 2 * hn;   // [L_* scalars (hn) | R_* scalars (hn)]
      const Fr* oR = oL + hn;
      reqs.push_back(req(cseg(0, (uint32_t)hn, d_ghi), none, oL));
      ci.push_back(cidx(p, sm.LA((int)j)));
      reqs.push_back(req(pseg(p, O_T, (uint32_t)hn, d_hi), none, oL));
      ci.push_back(cidx(p, sm.LT((int)j)));
      reqs.push_back(req(pseg(p, O_U, (uint32_t)hn, d_hi), none, oL));
      ci.push_back(cidx(p, sm.LU((int)j)));
      reqs.push_back(req(cseg(0, (uint32_t)hn, d_glo), none, oR));
      ci.push_back(cidx(p, sm.RA((int)j)));
      reqs.push_back(req(pseg(p, O_T, (uint32_t)hn, d_lo), none, oR));
      ci.push_back(cidx(p, sm.RT((int)j)));
      reqs.push_back(req(pseg(p, O_U, (uint32_t)hn, d_lo), none, oR));
      ci.push_back(cidx(p, sm.RU((int)j)));
    }
    dp.smsm.emplace_back(new TblPlan());
    if (dp.fused_smsm) {
      dp.smsm.back()->keep_order = true;
      dp.smsm.back()->combined = f_combine;
      dp.smsm.back()->force_fix_wpw = f_fix_smsm;
      dp.smsm.back()->force_tbl_wpw = f_tbl_smsm;
    }
    build_plan(*dp.smsm.back(), reqs, ci);
  }
  if (lt.on) {
    const int m = lt.m, hm = m / 2;
    const size_t BM = B * (size_t)m, ent = late_tab_entries();
    lt.jac.ensure((size_t)LATE_FAMILIES * BM);
    lt.tab.ensure((size_t)LATE_FAMILIES * BM * ent);
    lt.gb_cols = iGb;
    lt.h_col = iH;
    lt.part.ensure(6 * B * (size_t)opt_.late_slices);
    lt.extra.ensure(4 * B);
    CPX_HIP(hipMemsetAsync(lt.extra.p, 0, 4 * B * sizeof(TJac), stream_));   // all-zero = the identity: the outputs without a CRS term keep it (ordered before the prove's launches on this stream)
    lt.ipa.clear();
    lt.smsm.clear();
    lt.ipa_comp.clear();
    lt.smsm_comp.clear();
    std::vector<uint32_t> meta(2 * 6 * B);
    for (size_t g = 0; g < 6 * B; g++) {
      meta[g] = (uint32_t)g;      // request g = partial sum g
      meta[6 * B + g] = 1;
    }
    for (int r = 0; r < lt.nr; r++) {
      const int j = (int)lt.j0 + r;
      LateRound a{};
      a.m = m;
      a.half = m >> (r + 1);
      a.nproofs = (int)B;
      a.slices = (int)opt_.late_slices;
      a.tab = lt.tab.p;
      a.scal = d_rout_.p;
      // inner_product_argument.rs:150-163: L_C = <c_L, G_R> + beta <c_L, d_R> H, L_D = <d_R, G'_L>, R_C = <c_R, G_L> + beta <c_R, d_L> H,
      // R_D = <d_L, G'_R>; scalar rows as k_ipa_round_scalars lays them out for m active elements
      a.nout = 4;
      a.total = (int)(4 * B) * a.slices;
      a.scal_proof_stride = (uint32_t)(4 * hm + 2);
      a.o[0] = LateOut{LATE_F_G, 1, 0u};                          // (+ the H terms at scalar offsets hm and 3 hm + 1: exec_late_round)
      a.o[1] = LateOut{LATE_F_GP, 0, (uint32_t)(hm + 1)};
      a.o[2] = LateOut{LATE_F_G, 0, (uint32_t)(2 * hm + 1)};
      a.o[3] = LateOut{LATE_F_GP, 1, (uint32_t)(3 * hm + 2)};
      lt.ipa.push_back(a);
      lt.ipa_comp.push_back(meta.size());
      for (size_t p = 0; p < B; p++)
        for (int s : {sm.LC(j), sm.LD(j), sm.RC(j), sm.RD(j)}) meta.push_back(cidx(p, s));
      // same_multiscalar_argument.rs:104-112: L_A, L_T, L_U = <x_L, {G_b, T_b, U_b}_R>, R_* = <x_R, {..}_L>
      a.nout = 6;
      a.total = (int)(6 * B) * a.slices;
      a.scal_proof_stride = (uint32_t)(2 * hm);
      const int fam[3] = {LATE_F_GB, LATE_F_T, LATE_F_U};
      for (int q = 0; q < 3; q++) {
        a.o[q] = LateOut{fam[q], 1, 0u};
        a.o[3 + q] = LateOut{fam[q], 0, (uint32_t)hm};
      }
      lt.smsm.push_back(a);
      lt.smsm_comp.push_back(meta.size());
      for (size_t p = 0; p < B; p++)
        for (int s : {sm.LA(j), sm.LT(j), sm.LU(j), sm.RA(j), sm.RT(j), sm.RU(j)}) meta.push_back(cidx(p, s));
    }
    lt.meta.ensure(meta.size());
    CPX_HIP(hipMemcpy(lt.meta.p, meta.data(), meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  // -- side stream: R = a x vec_R, S = a x vec_S (curdleproofs.rs:112-113) and the four T_2 = s * {R, S} + r * H
  //    scalar multiplications (curdleproofs.rs:115-116, same_scalar_argument.rs:60-61)
  {
    std::vector<MsmTask> mt(2 * B);
    std::vector<uint32_t> md(2 * B);
    std::vector<SmulTask> st(4 * B);
    for (size_t p = 0; p < B; p++) {
      mt[2 * p] = MsmTask{pp(p), nullptr, d_veca_.p + p * ell, (uint32_t)ell, 0, (uint32_t)(2 * p * ell)};
      mt[2 * p + 1] = MsmTask{pp(p) + ell, nullptr, d_veca_.p + p * ell, (uint32_t)ell, 0, (uint32_t)((2 * p + 1) * ell)};
      md[2 * p] = slot_index(p, SL_R);
      md[2 * p + 1] = slot_index(p, SL_S);
      const Fr* kk = dp.k.p + p;
      const Fr* rk = rnd(p, ri.RK());
      st[4 * p + 0] = SmulTask{slot(p, sm.TMP(0)), slot(p, SL_R), slot(p, SL_CMT2), kk, 0, 0};
      st[4 * p + 1] = SmulTask{slot(p, sm.TMP(1)), slot(p, SL_S), slot(p, SL_CMU2), kk, 0, 0};
      st[4 * p + 2] = SmulTask{slot(p, sm.TMP(2)), slot(p, SL_R), slot(p, sm.CMA2()), rk, 0, 0};
      st[4 * p + 3] = SmulTask{slot(p, sm.TMP(3)), slot(p, SL_S), slot(p, sm.CMB2()), rk, 0, 0};
    }
    CPX_HIP(hipMemcpy(side_.tasks.p, mt.data(), mt.size() * sizeof(MsmTask), hipMemcpyHostToDevice));
    CPX_HIP(hipMemcpy(side_.dst.p, md.data(), md.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    CPX_HIP(hipMemcpy(side_.stasks.p, st.data(), st.size() * sizeof(SmulTask), hipMemcpyHostToDevice));
    dp.side_cols = idx_list({(uint32_t)SL_R, (uint32_t)SL_S, (uint32_t)SL_CMT2, (uint32_t)SL_CMU2, (uint32_t)sm.CMA2(), (uint32_t)sm.CMB2()});
  }
  if (!tab_.dstream) CPX_HIP(hipStreamCreateWithFlags(&tab_.dstream, hipStreamNonBlocking));
  if (!dp.ev_t1) {
    CPX_HIP(hipEventCreateWithFlags(&dp.ev_t1, hipEventDisableTiming));
    CPX_HIP(hipEventCreateWithFlags(&dp.ev_t2, hipEventDisableTiming));
    CPX_HIP(hipEventCreateWithFlags(&dp.ev_a2, hipEventDisableTiming));
  }
  if (!dp.ev_a) {
    CPX_HIP(hipEventCreateWithFlags(&dp.ev_a, hipEventDisableTiming));
    CPX_HIP(hipEventCreateWithFlags(&dp.ev_b, hipEventDisableTiming));
    CPX_HIP(hipEventCreateWithFlags(&dp.ev_c, hipEventDisableTiming));
    CPX_HIP(hipEventCreateWithFlags(&dp.ev_d, hipEventDisableTiming));
  }
  // the shared scratch of exec_plan sized for the largest phase now: nothing is allocated (no implicit device synchronisation)
  // between the first and the last launch of a prove
  {
    size_t max_parts = 1, max_sets = 1;
    auto upd = [&](const TblPlan& pl) {
      max_parts = std::max(max_parts, pl.nparts);
      max_sets = std::max(max_sets, pl.fix_sets + pl.tbl_sets);
    };
    upd(dp.p1);
    upd(dp.p1b);
    upd(dp.p1t);   // (in line on the main stream when the table stream is off: large batches, serial_streams)
    upd(dp.p2);
    upd(dp.p3);
    for (auto& pl : dp.ipa)
      if (pl) upd(*pl);
    for (auto& pl : dp.smsm)
      if (pl) upd(*pl);
    if (lt.on) max_parts = std::max(max_parts, 6 * B);
    if (dp.fused || dp.fused_smsm) {
      dp.rpart.ensure(max_parts);
      dp.rcount.ensure(B);
    }
    d_tpart_.ensure(max_parts);
    ensure_raw_sets(max_sets);
    d_tbltmp_.ensure(std::max(table_chunk_rows(B) * NP * (size_t)(copies_ / 2 - 1),   // one chunk of the table build ...
                              (3 * B * (size_t)lt.m + 63) / 64 * 64 * late_tmp_per_lane()));               // ... or the multiples of the late rounds' materialised points
  }
  dp.signature = sig;
}

// Whether the loaded batch runs its table build and phase 1t on the table stream (option table_stream_max, stated for n <= 256)
bool Engine::table_stream_on() const { return !opt_.serial_streams && (long)(B_ * std::max<size_t>(n_, 256)) <= opt_.table_stream_max * 256L; }

// The high-priority stream of a small batch's transcript prefix (created on first use; a plain stream where priorities are not offered)
hipStream_t Engine::prefix_stream() {
  if (!side_.hi_stream) {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) {
      (void)hipGetLastError();
      lo = hi = 0;
    }
    if (hipStreamCreateWithPriority(&side_.hi_stream, hipStreamNonBlocking, hi) != hipSuccess) {
      (void)hipGetLastError();
      CPX_HIP(hipStreamCreateWithFlags(&side_.hi_stream, hipStreamNonBlocking));
    }
  }
  return side_.hi_stream;
}

// The kernel launches of one device-resident prove (everything between the upload of the witnesses and the download of the proofs),
// on the main stream and — forked and joined through events — the side stream.  No host synchronisation, no allocation.
void Engine::enqueue_prove_device() {
  const size_t B = B_, ell = ell_, n = n_, L = L_;
  const SlotMap sm(L);
  const size_t NP = np(), NS = sm.count();
  DevProver& dp = dprove_;
  const ProveDev& d = dp.dev;
  const int Bi = (int)B;
  // (measurement aid, option serial_streams: the side stream's work in line on the main stream, so that every kernel's duration is its
  // own and a pass is the plain sum of its kernels)
  hipStream_t const side = opt_.serial_streams ? stream_ : side_.stream;
  // -- P0: compressed instance vectors, M -> affine (table source slot 0); side stream: transcript prefix (instance + M
  //    absorbed, vec_a drawn), then R and S; main stream: the per-proof tables
  tick("k_compress", 0, (double)(4 * ell * B));
  launch_compress(d_pp_.p, (int)(4 * ell), (int)pp_stride_, Bi, d_bytes_.p, stream_);
  tock();
  launch_finalize(d_Mjac_.p, Bi, d_psrc_.p, dp.mdst.p, d_mcomp_.p, stream_);
  CPX_HIP(hipEventRecord(dp.ev_a, stream_));
  CPX_HIP(hipStreamWaitEvent(side, dp.ev_a, 0));
  // (a small batch: the prefix's waves claim whole SIMDs and go through a high-priority queue, so that they are placed before the
  // waves of the table build that becomes ready at the same moment — otherwise they wait for SIMDs to drain)
  const bool excl = (long)B <= opt_.transcript_excl_max && !opt_.serial_streams;
  hipStream_t const pre = excl ? prefix_stream() : side;
  if (pre != side) {
    // head start: the main and the table stream continue behind an event the prefix stream records just before it launches — one
    // more hop between queues (~80 us) than the prefix kernel needs to place its waves on empty SIMDs
    CPX_HIP(hipStreamWaitEvent(pre, dp.ev_a, 0));
    CPX_HIP(hipEventRecord(dp.ev_a2, pre));
  }
  tick("k_transcript_step1", 0, (double)B);
  launch_transcript_step1(d_bytes_.p, d_mcomp_.p, Bi, (int)ell, d_tstate_.p, d_veca_.p, pre, (long)B * 256 >= opt_.transcript_lane_min_batch * (long)std::max<size_t>(n, 256), excl);
  tock();
  CPX_HIP(hipEventRecord(dp.ev_b, pre));
  if (pre != side) CPX_HIP(hipStreamWaitEvent(side, dp.ev_b, 0));
  {
    const int nt = 2 * Bi;
    // (R and S of a proof share the scalars vec_a: one wave per (proof, window) serves both, kernels.h launch_msm_endo_pairs)
    tick(opt_.rs_pairs ? "k_msm_tblw_pair" : "k_msm_tblw<2, true>", 128.0 * 2 * ell * B, (double)(2 * ell * B));
    if (opt_.rs_pairs) launch_msm_endo_pairs(side_.tasks.p, Bi, (int)ell, side_.conv.p, side_.digits.p, side_.ttasks.p, side_.raw.p, side_.rawslot.p, side);
    else launch_msm_endo(side_.tasks.p, nt, (int)ell, side_.conv.p, side_.digits.p, side_.ttasks.p, side_.raw.p, side_.rawslot.p, side);
    tock();
    if (opt_.serial_streams) tick("k_reduce_sets", 0, (double)(nt * 32), true);   // (a span is bracketed on the main stream)
    launch_reduce_sets(opt_, side_.raw.p, side_.rawslot.p, 0, nt * 32, side_.mid.p, side_.part.p, side, Bi);
    if (opt_.serial_streams) tock();
    tick("k_msm_tail", 0, (double)nt);
    launch_msm_tail(opt_, side_.part.p, nullptr, side_.res.p, nt, 16, 8, side, nullptr, 0, 2);
    tock();
    launch_finalize(side_.res.p, nt, d_pp_.p, side_.dst.p, nullptr, side);
  }
  // table stream: the per-proof tables, then B_t and B_u (phase 1t) — nothing on the main stream needs a table before phase 2 (M's row)
  // nor B_t, B_u before the SameMSM transcript step
  hipStream_t const tabs = table_stream_on() ? tab_.dstream : stream_;
  if (tabs != stream_) CPX_HIP(hipStreamWaitEvent(tabs, pre != side ? dp.ev_a2 : dp.ev_a, 0));
  if (pre != side) CPX_HIP(hipStreamWaitEvent(stream_, dp.ev_a2, 0));
  // (in equal chunks of table_chunk_rows(B) proofs, one after the other on this stream, through ONE chunk's worth of scratch: 6.7 instead of
  // 13.5 GB per context at 8192 proofs of ell = 252; a chunk of 4096 proofs is 2 M threads — the GPU is as full as with all rows at once)
  for (size_t r0 = 0, ch = table_chunk_rows(B); r0 < B; r0 += ch) {
    const size_t rows = std::min(ch, B - r0);
    tick("k_table_build", 0, (double)(rows * NP));
    launch_table_build(opt_, d_psrc_.p + r0 * NP, NP, d_ptab_.p + r0 * (size_t)copies_ * NP, (int)rows, (size_t)copies_ * NP, (int)NP, (int)NP, copies_, true, d_tbltmp_.p, tabs);
    tock();
  }
  CPX_HIP(hipEventRecord(dp.ev_t1, tabs));
  exec_plan(dp.p1t, dp.slotcomp.p, true);
  CPX_HIP(hipEventRecord(dp.ev_t2, tabs));
  if (opt_.p1_split) {   // the randomness-only commitments beside the prefix, A behind it
    exec_plan(dp.p1, dp.slotcomp.p);
    CPX_HIP(hipEventRecord(dp.ev_c, stream_));
    CPX_HIP(hipStreamWaitEvent(stream_, dp.ev_b, 0));
    launch_ps_aperm(d, Bi, stream_);
    exec_plan(dp.p1b, dp.slotcomp.p);
  } else {
    CPX_HIP(hipStreamWaitEvent(stream_, dp.ev_b, 0));   // vec_a and the transcript states
    // -- P1
    launch_ps_aperm(d, Bi, stream_);
    exec_plan(dp.p1, dp.slotcomp.p);
    CPX_HIP(hipEventRecord(dp.ev_c, stream_));           // the r * H points of the T_2 commitments are in TMP0..3
  }
  CPX_HIP(hipStreamWaitEvent(side, dp.ev_c, 0));
  tick("k_smul", 0, (double)(4 * B));
  launch_smul(side_.stasks.p, 4 * Bi, 1, side, /*exclusive_simd=*/side != stream_, opt_.serial_streams ? 0 : opt_.smul_quad_max);   // (a small batch: at most 16 waves, each on a SIMD of its own)
  tock();
  launch_compress_cols(d_pp_.p + 4 * ell, dp.side_cols, 6, (int)pp_stride_, Bi, dp.slotcomp.p, (int)NS, side);
  CPX_HIP(hipEventRecord(dp.ev_d, side));
  launch_ps_sameperm(d, Bi, stream_);
  // -- P2, P3
  if (tabs != stream_) CPX_HIP(hipStreamWaitEvent(stream_, dp.ev_t1, 0));   // M's table row
  exec_plan(dp.p2, dp.slotcomp.p);
  launch_ps_gprod(d, Bi, stream_);
  exec_plan(dp.p3, dp.slotcomp.p);
  launch_ps_ipa_setup(d, Bi, stream_);
  // -- IPA rounds (inner_product_argument.rs:150-186 in all-MSM form; the last log2(m) of a large batch on materialised folded
  //    bases: late.hip)
  const DevProver::Late& lt = dp.late;
  const size_t BM = B * (size_t)lt.m;
  // fused rounds (round.hip): the descriptor of a round's launch from its plan (the scratch pointers are read at enqueue time: a later
  // batch of another shape may have regrown them)
  if (dp.fused || dp.fused_smsm) CPX_HIP(hipMemsetAsync(dp.rcount.p, 0, B * sizeof(uint32_t), stream_));
  auto round_dev = [&](const TblPlan& pl, int nreq, bool last) {
    RoundDev rd{};
    rd.ftasks = pl.ftasks.p;
    rd.ttasks = pl.ttasks.p;
    rd.nf = (int)(pl.nft / B);
    rd.nt = (int)(pl.ntt / B);
    rd.fix_wpw = pl.fix_wpw;
    rd.tbl_wpw = rd.nt ? pl.tbl_wpw : 0;
    rd.combine = (rd.nt && pl.combined && rd.nf * (16 / pl.fix_wpw) == rd.nt * (32 / pl.tbl_wpw)) ? 1 : 0;
    rd.wpp = rd.combine ? (uint32_t)(rd.nt * (32 / pl.tbl_wpw)) : (uint32_t)(rd.nf * (16 / pl.fix_wpw) + (rd.nt ? rd.nt * (32 / pl.tbl_wpw) : 0));
    rd.nreq = nreq;
    rd.nproofs = Bi;
    rd.max_count = (uint32_t)std::max(16 / pl.fix_wpw, rd.nt ? 64 / pl.tbl_wpw : 0);
    rd.next_scalars = last ? 0 : 1;
    rd.fixtab = fixtab();
    rd.nc = (int)nc();
    rd.fraw = d_raw_.p;
    rd.fraw_slot = d_rawslot_.p;
    rd.traw = d_raw_.p + pl.fix_sets * raw_set_words();
    rd.traw_slot = d_rawslot_.p + pl.fix_sets;
    rd.part = dp.rpart.p;
    rd.meta = pl.meta.p;
    rd.comp_index = pl.meta.p + 3 * pl.nt;
    rd.counter = dp.rcount.p;
    rd.scal_out = d_rout_.p;
    return rd;
  };
  if (dp.fused) {
    launch_ipa_round_scalars(d_rvec_.p, Bi, (int)n, (int)(n >> 1), d_rbeta_.p, d_rout_.p, stream_);   // round 0's scalars; every later round's come from the tail before it
    for (size_t j = 0; j < L; j++) {
      const TblPlan& pl = *dp.ipa[j];
      tick("k_round_fused<ipa>", 128.0 * pl.pts_fix, pl.pts_fix);
      launch_round_fused(round_dev(pl, 4, j + 1 == L), d, (int)j, true, stream_);
      tock();
    }
  }
  for (size_t j = 0; j < L && !dp.fused; j++) {
    const int half = (int)(n >> (j + 1));
    if (lt.on && j == lt.j0) {   // G^(j0) and G'^(j0) from the table of multiples (fold coefficients S_G, S_G' = u o S^-1), their small multiples
      tick("k_late_fix", 128.0 * 2 * n * B, (double)(2 * n * B));
      launch_late_fix(d_rvec_.p + 2 * n, 4 * n, nullptr, (int)n, lt.m, fixtab(), fix_bits_, (int)nc(), lt.jac.p + LATE_F_G * BM, (size_t)lt.m, Bi, stream_);
      tock();
      launch_late_fix(d_rvec_.p + 3 * n, 4 * n, nullptr, (int)n, lt.m, fixtab(), fix_bits_, (int)nc(), lt.jac.p + LATE_F_GP * BM, (size_t)lt.m, Bi, stream_);
      tick("k_late_tables", 0, (double)(2 * BM));
      launch_late_tables(lt.jac.p + LATE_F_G * BM, lt.tab.p + LATE_F_G * BM * late_tab_entries(), d_tbltmp_.p, (int)(2 * BM), stream_);
      tock();
      launch_late_restart(d_rvec_.p, (int)n, lt.m, Bi, stream_);
    }
    if (lt.on && j >= lt.j0) {
      launch_ipa_round_scalars(d_rvec_.p, Bi, (int)n, half, d_rbeta_.p, d_rout_.p, stream_, lt.m);
      exec_late_round(lt.ipa[j - lt.j0], lt.ipa_comp[j - lt.j0], "k_late_msm");
      launch_ps_ipa_round(d, Bi, (int)j, stream_);
      launch_ipa_round_fold(d_rvec_.p, Bi, (int)n, half, d_rgam_.p, stream_, lt.m);
      continue;
    }
    launch_ipa_round_scalars(d_rvec_.p, Bi, (int)n, half, d_rbeta_.p, d_rout_.p, stream_);
    exec_plan(*dp.ipa[j], dp.slotcomp.p);
    launch_ps_ipa_round(d, Bi, (int)j, stream_);
    launch_ipa_round_fold(d_rvec_.p, Bi, (int)n, half, d_rgam_.p, stream_);
  }
  // -- SameScalar, SameMSM step 1 (needs R, S and the T_2 commitments of the side stream)
  CPX_HIP(hipStreamWaitEvent(stream_, dp.ev_d, 0));
  if (tabs != stream_) CPX_HIP(hipStreamWaitEvent(stream_, dp.ev_t2, 0));   // B_t, B_u in the slot registry
  launch_ps_smsm_setup(d, Bi, stream_);
  if (dp.fused_smsm) {
    launch_smsm_round_scalars(dp.rvec2.p, Bi, (int)n, (int)(n >> 1), d_rout_.p, stream_);
    for (size_t j = 0; j < L; j++) {
      const TblPlan& pl = *dp.smsm[j];
      tick("k_round_fused<smsm>", 128.0 * (pl.pts_fix + pl.pts_tbl), pl.pts_fix + pl.pts_tbl);
      launch_round_fused(round_dev(pl, 6, j + 1 == L), d, (int)j, false, stream_);
      tock();
    }
  }
  for (size_t j = 0; j < L && !dp.fused_smsm; j++) {
    const int half = (int)(n >> (j + 1));
    if (lt.on && j == lt.j0) {   // T_b^(j0), U_b^(j0), G_b^(j0): one shared digit sequence per proof (fold coefficients S_M)
      // (T_b, U_b with two lanes per output fill a wave at m = 16, with one lane per output at m = 32; m = 64: two waves per proof; G_b comes
      // from the table of multiples like G and G')
      tick("k_late_uniform", 128.0 * 2 * n * B, (double)(2 * n * B));
      launch_late_uniform(dp.rvec2.p + n, 2 * n, d_ptab_.p, (size_t)copies_ * NP, (int)NP, ctab(), (int)nc(), lt.gb_cols, (int)n, lt.m, 2, 4 * lt.m <= 64 ? 2 : 1,
                          lt.jac.p + LATE_F_T * BM, BM, Bi, stream_);
      tock();
      launch_late_fix(dp.rvec2.p + n, 2 * n, lt.gb_cols, (int)n, lt.m, fixtab(), fix_bits_, (int)nc(), lt.jac.p + LATE_F_GB * BM, (size_t)lt.m, Bi, stream_);
      tick("k_late_tables", 0, (double)(3 * BM));
      launch_late_tables(lt.jac.p + LATE_F_T * BM, lt.tab.p + LATE_F_T * BM * late_tab_entries(), d_tbltmp_.p, (int)(3 * BM), stream_);
      tock();
    }
    if (lt.on && j >= lt.j0) {
      launch_smsm_round_scalars(dp.rvec2.p, Bi, (int)n, half, d_rout_.p, stream_, lt.m);
      exec_late_round(lt.smsm[j - lt.j0], lt.smsm_comp[j - lt.j0], "k_late_msm");
      launch_ps_smsm_round(d, Bi, (int)j, stream_);
      launch_smsm_round_fold(dp.rvec2.p, Bi, (int)n, half, d_rgam_.p, stream_, lt.m);
      continue;
    }
    launch_smsm_round_scalars(dp.rvec2.p, Bi, (int)n, half, d_rout_.p, stream_);
    exec_plan(*dp.smsm[j], dp.slotcomp.p);
    launch_ps_smsm_round(d, Bi, (int)j, stream_);
    launch_smsm_round_fold(dp.rvec2.p, Bi, (int)n, half, d_rgam_.p, stream_);
  }
  launch_ps_serialize(d, Bi, stream_);
}

void Engine::batch_prove_device(const uint32_t* permutation, const uint8_t* k_in, const uint8_t* m_blinders, const uint8_t* rand, uint8_t* proofs_out) {
  HostSpan wall(this, "host_prove_wall");
  CPX_HIP(hipSetDevice(device_));
  const size_t B = B_, ell = ell_, n = n_, nrand = 3 * n + 9;
  prepare_device_prover();
  DevProver& dp = dprove_;

  // witnesses and the prover's random draws: the only host -> device traffic of a prove
  CPX_HIP(hipMemcpyAsync(dp.perm.p, permutation, B * ell * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(dp.k.p, k_in, B * sizeof(Fr), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(dp.mbl.p, m_blinders, B * 4 * sizeof(Fr), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(dp.rnd.p, rand, B * nrand * sizeof(Fr), hipMemcpyHostToDevice, stream_));

  enqueue_prove_device();
  CPX_HIP(hipMemcpyAsync(proofs_out, dp.proofs.p, B * proof_size(), hipMemcpyDeviceToHost, stream_));
  wait_stream_blocking();
  flush_timers();
  if (opt_.trace) {   // debugging aid: the challenges and responses of proof 0, same format as the host-driven prover prints
    std::vector<Fr> sc(SC_COUNT);
    CPX_HIP(hipMemcpy(sc.data(), dp.sc.p, SC_COUNT * sizeof(Fr), hipMemcpyDeviceToHost));
    const char* names[SC_COUNT] = {"beta_sp", "alpha_sp", "-beta_g_inv", "alpha_g", "gprod", "beta_g", "beta_g_inv", "r_p", "z_ip", "alpha_i", "beta_i",
                                   "alpha_s", "alpha_m", "z_k", "z_t", "z_u", "c_final", "d_final", "x_final"};
    for (int i = 0; i < SC_COUNT; i++)
      if (names[i]) trace_scalar(names[i], sc[i]);
  }
}

// ---------------------------------------------------------------- verifier
void Engine::prepare_device_verifier(size_t rand_stride) {
  const size_t B = B_, ell = ell_, n = n_, L = L_;
  const SlotMap sm(L);
  const size_t NS = sm.count(), NPP = sm.n_proof_points(), NM = 6 + NPP, NI = 4 * ell, NPT = NI + NM, psz = proof_size();
  DevVerifier& dv = dverify_;
  d_bytes_.ensure(B * 4 * ell * 48);
  d_mcomp_.ensure(B * 48);
  d_tstate_.ensure(B * 27);
  d_veca_.ensure(B * ell);
  dv.proofs.ensure(B * psz);
  dv.rnd.ensure(B * 12);
  dv.vsc.ensure(B * (size_t)VSC_COUNT);
  dv.slotcomp.ensure(B * NS * 48);
  dv.status.ensure(B * NPP);
  dv.scal.ensure(B * NPT + n);   // + the summed CRS scalars of a fused batch
  dv.scal_crs.ensure(B * n);
  dv.flags.ensure(B);
  dv.src_off.ensure(B * NPP);
  dv.dst.ensure(B * NPP);
  dv.mdst.ensure(B);
  dv.mtasks.ensure(B);
  dv.ftasks.ensure(B + 1);
  dv.gidx.ensure(B * NPT);
  d_conv_.ensure(2 * B * NPT);
  d_digits_.ensure(9 * B * NPT);
  d_ttasks_.ensure(B);
  d_part_.ensure(B * 32);
  d_res_.ensure(B);
  d_comp_.ensure(B * 48);
  const std::vector<const void*> sig = {d_pp_.p,       d_bytes_.p,  d_mcomp_.p,     d_tstate_.p, d_veca_.p,  dv.proofs.p, dv.rnd.p,    dv.vsc.p, dv.slotcomp.p,
                                        dv.status.p,   dv.scal.p,   dv.scal_crs.p,  dv.flags.p,  dv.mtasks.p, dv.ftasks.p, dv.gidx.p,  ctab(),   fixtab(),
                                        (const void*)(uintptr_t)B, (const void*)(uintptr_t)ell, (const void*)(uintptr_t)fix_bits_};
  dv.dev.rand_stride = (int)rand_stride;
  if (sig == dv.signature) return;
  dv.signature.clear();
  VerifyDev& d = dv.dev;
  d.ell = (int)ell;
  d.n = (int)n;
  d.L = (int)L;
  d.NS = (int)NS;
  d.NM = (int)NM;
  d.psz = psz;
  d.proofs = dv.proofs.p;
  d.rnd = dv.rnd.p;
  d.tstate = d_tstate_.p;
  d.veca = d_veca_.p;
  d.vsc = dv.vsc.p;
  d.slotcomp = dv.slotcomp.p;
  d.inst_comp = d_bytes_.p;
  d.mcomp = d_mcomp_.p;
  d.status = dv.status.p;
  d.scal = dv.scal.p;
  d.scal_crs = dv.scal_crs.p;
  d.flags = dv.flags.p;
  memcpy(d.crs_h_comp, crs_H_comp_, 48);

  // where the proof points sit inside a serialized proof (slot order == serialisation order, the seven scalars interleaved)
  std::vector<uint32_t> pt_off(NPP);
  for (size_t q = 0; q < NPP; q++) pt_off[q] = (uint32_t)(48 * q + (q >= 9 ? 32 : 0) + (q >= 11 + 4 * L ? 64 : 0) + (q >= 15 + 4 * L ? 96 : 0));
  std::vector<uint32_t> so(B * NPP), ds(B * NPP), md(B);
  for (size_t p = 0; p < B; p++) {
    for (size_t q = 0; q < NPP; q++) {
      so[p * NPP + q] = (uint32_t)(p * psz + pt_off[q]);
      ds[p * NPP + q] = slot_index(p, SL_A + (int)q);
    }
    md[p] = slot_index(p, SL_M);
  }
  CPX_HIP(hipMemcpy(dv.src_off.p, so.data(), so.size() * 4, hipMemcpyHostToDevice));
  CPX_HIP(hipMemcpy(dv.dst.p, ds.data(), ds.size() * 4, hipMemcpyHostToDevice));
  CPX_HIP(hipMemcpy(dv.mdst.p, md.data(), md.size() * 4, hipMemcpyHostToDevice));
  // D = B - beta^-1 sum(G) + alpha sum(H) (grand_product_argument.rs:223) and A' = A + cm_T.T_1 + cm_U.T_1 (curdleproofs.rs:258)
  // are hashed into the transcript, so they are needed as bytes
  {
    const TblSeg none{nullptr, nullptr, 0, 0};
    std::vector<TblReq> reqs;
    std::vector<uint32_t> ci;
    for (size_t p = 0; p < B; p++) {
      TblReq rd{cseg(n + 3, 2), nullptr, none, nullptr, slot_index(p, sm.D())};
      rd.dev = dv.vsc.p + p * VSC_COUNT + VSC_NEG_BETA_G_INV;
      rd.add[0] = slot_index(p, SL_B);
      TblReq ra{none, nullptr, none, nullptr, slot_index(p, sm.APRIME())};
      ra.add[0] = slot_index(p, SL_A);
      ra.add[1] = slot_index(p, SL_CMT1);
      ra.add[2] = slot_index(p, SL_CMU1);
      reqs.push_back(rd);
      ci.push_back((uint32_t)(p * NS + sm.D()));
      reqs.push_back(ra);
      ci.push_back((uint32_t)(p * NS + sm.APRIME()));
    }
    build_plan(dv.pd, reqs, ci);
  }
  // the accumulated check: per proof one fixed-base task over G | Hvec and one bucket MSM over R | S | T | U and the misc slots
  std::vector<uint32_t> all_idx(NPT);   // row-relative gather list
  for (size_t i = 0; i < NI; i++) all_idx[i] = (uint32_t)i;
  for (size_t j = 0; j < NM; j++) all_idx[NI + j] = (uint32_t)(NI + j);   // misc index j == slot j (SL_H .. SL_M, then the proof points)
  const uint32_t* d_all = idx_list(all_idx);
  dv.fix_wpw = msm_fix_windows_per_wave(opt_, (int)B, fix_bits_);
  dv.fix_parts = msm_fix_parts(fix_bits_, dv.fix_wpw);
  dv.fix_wpw1 = msm_fix_windows_per_wave(opt_, 1, fix_bits_);
  dv.fix_parts1 = msm_fix_parts(fix_bits_, dv.fix_wpw1);
  std::vector<MsmTask> mt(B);
  std::vector<FixTask> ft(B + 1);
  std::vector<uint32_t> gi(B * NPT);
  for (size_t p = 0; p < B; p++) {
    mt[p] = MsmTask{pp(p), d_all, dv.scal.p + p * NPT, (uint32_t)NPT, 0, (uint32_t)(p * NPT)};
    ft[p] = FixTask{nullptr, dv.scal_crs.p + p * n, 0, (uint32_t)n, 0, (uint32_t)(p * dv.fix_parts)};
    for (size_t i = 0; i < NPT; i++) gi[p * NPT + i] = (uint32_t)(p * pp_stride_ + i);
  }
  ft[B] = FixTask{nullptr, dv.scal.p + B * NPT, 0, (uint32_t)n, 0, 0};   // fused batch: the CRS scalars summed over the proofs
  CPX_HIP(hipMemcpy(dv.mtasks.p, mt.data(), B * sizeof(MsmTask), hipMemcpyHostToDevice));
  CPX_HIP(hipMemcpy(dv.ftasks.p, ft.data(), (B + 1) * sizeof(FixTask), hipMemcpyHostToDevice));
  CPX_HIP(hipMemcpy(dv.gidx.p, gi.data(), gi.size() * 4, hipMemcpyHostToDevice));
  // fused batch: the proofs in up to 256 groups, every group one task of the endomorphism bucket-list kernel
  dv.G = (B + 255) / 256;
  dv.NT = (B + dv.G - 1) / dv.G;
  dv.gtasks.ensure(dv.NT);
  std::vector<MsmTask> gt(dv.NT);
  for (size_t g = 0; g < dv.NT; g++) {
    const size_t off = g * dv.G * NPT, np_ = std::min(dv.G, B - g * dv.G) * NPT;
    gt[g] = MsmTask{d_pp_.p, dv.gidx.p + off, dv.scal.p + off, (uint32_t)np_, 0, (uint32_t)off};
  }
  CPX_HIP(hipMemcpy(dv.gtasks.p, gt.data(), dv.NT * sizeof(MsmTask), hipMemcpyHostToDevice));
  if (!dv.ev_a) {
    CPX_HIP(hipEventCreateWithFlags(&dv.ev_a, hipEventDisableTiming));
    CPX_HIP(hipEventCreateWithFlags(&dv.ev_b, hipEventDisableTiming));
  }
  dv.signature = sig;
}

// curdleproofs.rs:197 for every loaded instance (verdict != nullptr) or BASELINE config 5 (fused_partial != nullptr), all on the
// device: the host uploads proofs and random factors, enqueues, and reads back 48 bytes + a flag word per proof (or one point).
void Engine::verify_core_device(const uint8_t* proofs, const uint8_t* rand, size_t rand_stride, int* verdict, uint8_t* fused_partial, int* fused_invalid) {
  CPX_HIP(hipSetDevice(device_));
  const size_t B = B_, ell = ell_, n = n_, L = L_;
  const SlotMap sm(L);
  const size_t NPP = sm.n_proof_points(), NM = 6 + NPP, NPT = 4 * ell + NM, psz = proof_size();
  prepare_device_verifier(rand_stride);
  DevVerifier& dv = dverify_;
  const VerifyDev& d = dv.dev;
  const int Bi = (int)B;
  CPX_HIP(hipMemcpyAsync(dv.proofs.p, proofs, B * psz, hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(dv.rnd.p, rand, B * rand_stride * sizeof(Fr), hipMemcpyHostToDevice, stream_));
  // -- V0: compressed instance vectors, M -> affine; side stream: transcript prefix; main: proof points -> slots
  tick("k_compress", 0, (double)(4 * ell * B));
  launch_compress(d_pp_.p, (int)(4 * ell), (int)pp_stride_, Bi, d_bytes_.p, stream_);
  tock();
  launch_finalize(d_Mjac_.p, Bi, d_pp_.p, dv.mdst.p, d_mcomp_.p, stream_);
  CPX_HIP(hipEventRecord(dv.ev_a, stream_));
  hipStream_t const side = opt_.serial_streams ? stream_ : side_.stream;
  CPX_HIP(hipStreamWaitEvent(side, dv.ev_a, 0));
  tick("k_transcript_step1", 0, (double)B);
  launch_transcript_step1(d_bytes_.p, d_mcomp_.p, Bi, (int)ell, d_tstate_.p, d_veca_.p, side, (long)B * 256 >= opt_.transcript_lane_min_batch * (long)std::max<size_t>(n, 256), (long)B <= opt_.transcript_excl_max && !opt_.serial_streams);
  tock();
  CPX_HIP(hipEventRecord(dv.ev_b, side));
  tick("k_decompress", 0, (double)(B * NPP));
  launch_decompress(opt_, dv.proofs.p, (int)(B * NPP), d_pp_.p, dv.dst.p, dv.status.p, 1, stream_, dv.src_off.p);
  tock();
  CPX_HIP(hipStreamWaitEvent(stream_, dv.ev_b, 0));
  // -- V1: transcript up to the grand-product beta; D and A' as bytes; the rest of the transcript and the scalars
  launch_vs_prefix(d, Bi, stream_);
  exec_plan(dv.pd, dv.slotcomp.p);
  launch_vs_scalars(d, Bi, stream_);
  // -- V2: the accumulated check(s)
  h_comp_.ensure(std::max<size_t>(B * 48, sizeof(Jac)));
  h_u32_.ensure(B);
  if (fused_partial) {
    launch_vs_crs_sum(dv.scal_crs.p, Bi, (int)n, dv.scal.p + B * NPT, stream_);
    d_tpart_.ensure(dv.fix_parts1);
    ensure_raw_sets(std::max<size_t>(dv.NT * 32, dv.fix_parts1));
    launch_msm_fix(dv.ftasks.p + B, 1, fixtab(), fix_bits_, dv.fix_wpw1, (int)nc(), d_raw_.p, d_rawslot_.p, stream_);
    reduce_sets(dv.fix_parts1, 0);
    const size_t NT = dv.NT, NT16 = (NT + 15) / 16 * 16, N = B * NPT;
    d_part_.ensure(NT * 32);
    d_wsum_.ensure(NT16 + NT16 / 16);
    d_ttasks_.ensure(NT);
    CPX_HIP(hipMemsetAsync(d_wsum_.p, 0, (NT16 + NT16 / 16) * sizeof(TJac), stream_));   // all-zero = identity: pads the summation levels
    tick("k_msm_tblw<2, true>", 128.0 * N, (double)N);
    launch_msm_endo(dv.gtasks.p, (int)NT, (int)(dv.G * NPT), d_conv_.p, d_digits_.p, d_ttasks_.p, d_raw_.p, d_rawslot_.p, stream_);
    tock();
    reduce_sets(0, NT * 32, d_part_.p);
    tick("k_msm_tail", 0, (double)NT, true);
    launch_msm_tail(opt_, d_part_.p, d_wsum_.p, nullptr, (int)NT, 16, 8, stream_, nullptr, 0, 2);                               // windows of a group
    launch_msm_tail(opt_, d_wsum_.p, d_wsum_.p + NT16, nullptr, (int)(NT16 / 16), 16, 0, stream_);                              // 16 groups each
    launch_msm_tail(opt_, d_wsum_.p + NT16, nullptr, d_res_.p, 1, (int)(NT16 / 16), 0, stream_, d_tpart_.p, dv.fix_parts1);      // + the fixed-base part
    tock();
    CPX_HIP(hipMemcpyAsync(h_comp_.p, d_res_.p, sizeof(Jac), hipMemcpyDeviceToHost, stream_));
    CPX_HIP(hipMemcpyAsync(h_u32_.p, dv.flags.p, B * 4, hipMemcpyDeviceToHost, stream_));
    wait_stream_blocking();
    memcpy(fused_partial, h_comp_.p, sizeof(Jac));
    int invalid = 0;
    for (size_t p = 0; p < B; p++) invalid += h_u32_.p[p] ? 1 : 0;
    if (fused_invalid) *fused_invalid = invalid;
    flush_timers();
    return;
  }
  d_tpart_.ensure(B * (size_t)dv.fix_parts);
  ensure_raw_sets(B * std::max<size_t>(dv.fix_parts, 32));
  tick(fix_kernel_name(fix_bits_, dv.fix_wpw), 128.0 * n * B, (double)(n * B));
  launch_msm_fix(dv.ftasks.p, Bi, fixtab(), fix_bits_, dv.fix_wpw, (int)nc(), d_raw_.p, d_rawslot_.p, stream_);
  tock();
  reduce_sets(B * dv.fix_parts, 0);
  tick("k_msm_tblw<2, true>", 128.0 * NPT * B, (double)(NPT * B));
  launch_msm_endo(dv.mtasks.p, Bi, (int)NPT, d_conv_.p, d_digits_.p, d_ttasks_.p, d_raw_.p, d_rawslot_.p, stream_);
  tock();
  reduce_sets(0, B * 32, d_part_.p);
  tick("k_msm_tail", 0, (double)B);
  launch_msm_tail(opt_, d_part_.p, nullptr, d_res_.p, Bi, 16, 8, stream_, d_tpart_.p, dv.fix_parts, 2);
  tock();
  tick("k_finalize", 0, (double)B);
  launch_finalize(d_res_.p, Bi, nullptr, nullptr, d_comp_.p, stream_);
  tock();
  CPX_HIP(hipMemcpyAsync(h_comp_.p, d_comp_.p, B * 48, hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipMemcpyAsync(h_u32_.p, dv.flags.p, B * 4, hipMemcpyDeviceToHost, stream_));
  wait_stream_blocking();
  for (size_t p = 0; p < B; p++) {
    const uint32_t f = h_u32_.p[p];
    const bool ok = h_comp_.p[p * 48] == 0xc0;
    verdict[p] = (f & 1u) ? CPX_ERR_DESERIALIZE : (((f & 2u) || !ok) ? CPX_ERR_VERIFY : CPX_OK);
  }
  flush_timers();
}

void trace_scalar(const char* name, const Fr& x) {
  fprintf(stderr, "[cpx trace] %-12s ", name);
  for (int i = 7; i >= 0; i--) fprintf(stderr, "%08x", x.v[i]);
  fprintf(stderr, "\n");
}

}  // namespace cpx
