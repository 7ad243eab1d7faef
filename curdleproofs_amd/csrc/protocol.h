// Device-resident protocol steps (protocol.hip): the Fiat-Shamir transcript and the scalar-field algebra between the MSM
// phases of CurdleproofsProof::{new,verify}, one wave per proof — shared declarations of the kernels and the host engine.
//
// With these kernels a whole batch advances through the protocol without the host: every phase's MSM requests read
// their scalars from device memory, the finalisation kernels leave the compressed results in a per-proof slot registry
// and the next step kernel hashes them and derives the next scalars.  Mirrors /root/reference/src/curdleproofs.rs:59-298,
// same_permutation_argument.rs, grand_product_argument.rs, inner_product_argument.rs, same_scalar_argument.rs,
// same_multiscalar_argument.rs (cited at the individual steps in protocol.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "mont32.hpp"

namespace cpx {

// ---- per-proof point registry ("slots") that follows the 4*ell instance points in d_pp_ ----
// CRS singles, M, then every proof point in serialisation order (curdleproofs.rs:300-310), then scratch.
enum { SL_H = 0, SL_GT, SL_GU, SL_GSUM, SL_HSUM, SL_M, SL_A, SL_CMT1, SL_CMT2, SL_CMU1, SL_CMU2, SL_R, SL_S, SL_B, SL_C, SL_BC, SL_BD, SL_IPA0 };
struct SlotMap {
  int L;
  CPX_HD explicit SlotMap(size_t l) : L((int)l) {}
  CPX_HD int LC(int j) const { return SL_IPA0 + j; }
  CPX_HD int RC(int j) const { return SL_IPA0 + L + j; }
  CPX_HD int LD(int j) const { return SL_IPA0 + 2 * L + j; }
  CPX_HD int RD(int j) const { return SL_IPA0 + 3 * L + j; }
  CPX_HD int CMA1() const { return SL_IPA0 + 4 * L; }
  CPX_HD int CMA2() const { return CMA1() + 1; }
  CPX_HD int CMB1() const { return CMA1() + 2; }
  CPX_HD int CMB2() const { return CMA1() + 3; }
  CPX_HD int BA() const { return CMA1() + 4; }
  CPX_HD int BT() const { return CMA1() + 5; }
  CPX_HD int BU() const { return CMA1() + 6; }
  CPX_HD int LA(int j) const { return CMA1() + 7 + j; }
  CPX_HD int LT(int j) const { return CMA1() + 7 + L + j; }
  CPX_HD int LU(int j) const { return CMA1() + 7 + 2 * L + j; }
  CPX_HD int RA(int j) const { return CMA1() + 7 + 3 * L + j; }
  CPX_HD int RT(int j) const { return CMA1() + 7 + 4 * L + j; }
  CPX_HD int RU(int j) const { return CMA1() + 7 + 5 * L + j; }
  CPX_HD int D() const { return CMA1() + 7 + 6 * L; }
  CPX_HD int APRIME() const { return D() + 1; }
  CPX_HD int TMP(int i) const { return D() + 2 + i; }   // 8 scratch results
  CPX_HD int count() const { return D() + 2 + 8; }
  CPX_HD int first_proof_point() const { return SL_A; }
  CPX_HD int n_proof_points() const { return 18 + 10 * L; }   // SL_A .. RU(L-1), contiguous
};

// ---- the prover's 3n+9 random draws (SURVEY 8b RNG contract), as indices into a proof's `rand` row ----
struct RandIdx {
  int n;
  CPX_HD explicit RandIdx(int n_) : n(n_) {}
  CPX_HD int AB() const { return 0; }              // vec_a_blinders[2]     curdleproofs.rs:86
  CPX_HD int CB() const { return 2; }              // vec_c_blinders[4]     grand_product_argument.rs:75
  CPX_HD int IR() const { return 6; }              // IPA r[n]              inner_product_argument.rs:46
  CPX_HD int IZ() const { return 6 + n; }          // IPA z[n-2]            inner_product_argument.rs:47
  CPX_HD int RT() const { return 2 * n + 4; }      // r_t, r_u              curdleproofs.rs:110-111
  CPX_HD int RU() const { return 2 * n + 5; }
  CPX_HD int RA() const { return 2 * n + 6; }      // r_a, r_b, r_k         same_scalar_argument.rs:56-58
  CPX_HD int RB() const { return 2 * n + 7; }
  CPX_HD int RK() const { return 2 * n + 8; }
  CPX_HD int VR() const { return 2 * n + 9; }      // SameMSM vec_r[n]      same_multiscalar_argument.rs:78
  CPX_HD int count() const { return 3 * n + 9; }
};

// ---- per-proof device state of the device-resident prover ----
// small scalars (Montgomery form), sc[p][SC_COUNT]; neighbours that one MSM request reads as a scalar vector are adjacent
enum {
  SC_BETA_SP = 0, SC_ALPHA_SP,          // B = A + alpha M + beta sum(G): request scalars [beta | alpha]
  SC_NEG_BETA_G_INV, SC_ALPHA_G,        // D = B - beta^-1 sum(G) + alpha sum(H): request scalars [-beta^-1 | alpha]
  SC_GPROD, SC_BETA_G, SC_BETA_G_INV, SC_RP, SC_ZIP, SC_ALPHA_I, SC_BETA_I, SC_ALPHA_S, SC_ALPHA_M,
  SC_ZK, SC_ZT, SC_ZU, SC_CFIN, SC_DFIN, SC_XFIN, SC_COUNT = 24
};
// vectors of n scalars, vec[p][V_COUNT][n]
enum { V_APERM = 0, V_FACT, V_C, V_D, V_U, V_ZZ, V_ZZU, V_COUNT };

struct ProveDev {
  int ell, n, L, NS;              // NS = SlotMap(L).count()
  size_t psz;                     // proof size in bytes
  const uint32_t* perm;           // [B][ell]
  const Fr* k;                    // [B]
  const Fr* mbl;                  // [B][4]   vec_m_blinders
  const Fr* rnd;                  // [B][3n+9]
  uint64_t* tstate;               // [B][27]  transcript state (25 lanes, pos, pos_begin)
  const Fr* veca;                 // [B][ell]
  Fr* vec;                        // [B][V_COUNT][n]
  Fr* sc;                         // [B][SC_COUNT]
  uint8_t* slotcomp;              // [B][NS][48]  compressed bytes of every slot
  const uint8_t* inst_comp;       // [B][4 ell][48]  compressed R | S | T | U
  const uint8_t* mcomp;           // [B][48]
  Fr* rvec;                       // [B][4][n]  IPA round vectors: c | d | SG | SGp
  Fr* rvec2;                      // [B][2][n]  SameMSM round vectors: x | SM
  Fr* rgam;                       // [B][2]     gamma, gamma^-1
  Fr* rbeta;                      // [B]        IPA beta
  uint8_t* proofs;                // [B][psz]
  uint8_t crs_h_comp[48];         // compressed H (the blinder slots of vec_T / vec_U in the SameMSM transcript)
};

// ---- per-proof device state of the device-resident verifier ----
enum {   // vsc[p][VSC_COUNT]
  VSC_NEG_BETA_G_INV = 0, VSC_ALPHA_G,   // D request scalars
  VSC_ALPHA_SP, VSC_BETA_SP, VSC_GPROD, VSC_BETA_G, VSC_BETA_G_INV, VSC_RP, VSC_CFIN, VSC_DFIN, VSC_ZK, VSC_ZT, VSC_ZU, VSC_XFIN, VSC_COUNT = 16
};
struct VerifyDev {
  int ell, n, L, NS, NM;          // NM = misc points of the accumulated check (6 + 18 + 10 L)
  size_t psz;
  int rand_stride;                // 8 (per-proof verdicts) or 12 (fused batch: four extra weights for the SameScalar relations)
  const uint8_t* proofs;          // [B][psz]
  const Fr* rnd;                  // [B][rand_stride]
  uint64_t* tstate;               // [B][27]
  const Fr* veca;                 // [B][ell]
  Fr* vsc;                        // [B][VSC_COUNT]
  uint8_t* slotcomp;              // [B][NS][48]: D and A' (computed on the device) land here
  const uint8_t* inst_comp;       // [B][4 ell][48]
  const uint8_t* mcomp;           // [B][48]
  const uint8_t* status;          // [B][n_proof_points]  decompression status of the proof points
  Fr* scal;                       // [B][4 ell + NM]: instance part | misc part of the accumulated check
  Fr* scal_crs;                   // [B][n]: its CRS part (G | Hvec)
  uint32_t* flags;                // [B]: bit 0 = undecodable (bad scalar / point encoding), bit 1 = structural rejection
  uint8_t crs_h_comp[48];
};

// ---- launchers (protocol.hip); B proofs, one wave each ----
void launch_ps_aperm(const ProveDev& d, int B, hipStream_t s);
void launch_ps_sameperm(const ProveDev& d, int B, hipStream_t s);
void launch_ps_gprod(const ProveDev& d, int B, hipStream_t s);
void launch_ps_ipa_setup(const ProveDev& d, int B, hipStream_t s);
void launch_ps_ipa_round(const ProveDev& d, int B, int j, hipStream_t s);
void launch_ps_smsm_setup(const ProveDev& d, int B, hipStream_t s);
void launch_ps_smsm_round(const ProveDev& d, int B, int j, hipStream_t s);
void launch_ps_serialize(const ProveDev& d, int B, hipStream_t s);
void launch_vs_prefix(const VerifyDev& d, int B, hipStream_t s);
void launch_vs_scalars(const VerifyDev& d, int B, hipStream_t s);
// out[i] = sum_p scal_crs[p * n + i], i < n  (the CRS scalars of a fused batch)
void launch_vs_crs_sum(const Fr* d_scal_crs, int B, int n, Fr* d_out, hipStream_t s);

}  // namespace cpx
