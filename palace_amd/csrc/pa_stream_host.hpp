// Host-side construction of the streaming kernel's index arrays and of the run form of the transpose map
// (pa_nd_hex_stream.hip).  Plain C++ without HIP so that tests/test_stream_host.py can compile it with g++ and check
// the encodings against a CPU model of the kernels' decode paths.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <vector>

namespace pa {

constexpr int kEssBit = 1 << 30;   // flag in a gather index: read this dof as zero
constexpr int kExclBit = 1 << 29;  // flag in the streaming kernel's index word: only copy of its dof

namespace streamhost {

struct RunHdr {
  // bits 0-26 first dof of the run, bits 27-30 its length - 1, bit 31 essential row (ParOperator's fix-up fused into the gather);
  // first entry of its copies in rpos (the next header's ptr ends them)
  uint32_t dof0;
  int32_t ptr;
};
constexpr uint32_t kRunDofMask = (1u << 27) - 1u;
inline int run_dof0(const RunHdr &h) { return (int)(h.dof0 & kRunDofMask); }
inline int run_len(const RunHdr &h) { return (int)((h.dof0 >> 27) & 15u) + 1; }
inline bool run_ess(const RunHdr &h) { return (h.dof0 >> 31) != 0; }

inline int dof_of(int32_t s) { return s >= 0 ? s : -1 - s; }

// Compressed element -> dof index of the streaming kernel.  The P entries of an element are sorted by dof, so they fall
// into a few runs of consecutive dofs (one per mesh entity: 12 edges, 6 faces and the interior of a hexahedron give at
// most 19); instead of one word per entry (576 B at p = 3) an element stores kIdxWords = 32 words (128 B):
//   word r < npl        entries 16 r .. 16 r + 15:  bits 0-15  bit j set = entry 16 r + j starts a run
//                                                     bits 16-20 number of runs that start before entry 16 r
//                                                     bits 21-28 position of the last run start before entry 16 r
//   word start0 + k     first dof of run k (k < kIdxWords - start0)
// dof(m) = start[run(m)] + m - first entry of run(m), see index_dof.  Signs, the only-copy flag and the essential flag
// travel in the flag word of pp.  start0 = kIdxStart0 (20 runs) for H(curl) elements (up to 9 slice words); H1 elements
// (at most 4 slice words, but 27 entities: 8 vertices + 12 edges + 6 faces + interior) use kIdxStart0H1 (28 runs).
constexpr int kIdxWords = 32, kIdxStart0 = 12, kIdxMaxRuns = kIdxWords - kIdxStart0, kIdxStart0H1 = 4;
// H(curl) blocks (at most 9 slice words before the run starts at word 12): word 11 names the element's entry in the dictionary
// of sorted -> tensor-order slot patterns (pa_nd_hex_stream.hip: build_stream)
constexpr int kIdxPattern = 11;
// flag words (four-point H(curl) kernel): bits 0-17 flip / exclusive, 18-26 essential; bit 31: the element's batch is affine
// (QData::batch_aff: the kernel reads the compact D of QData::d_aff)
constexpr uint32_t kAffBit = 1u << 31;

inline int index_dof(const uint32_t *ic, int m, int start0 = kIdxStart0) {  // host model of the device decode (gather lambdas)
  const int r = m >> 4, t = m & 15;
  const uint32_t w = ic[r], low = (w & 0xffffu) & ((2u << t) - 1u);
  int bits = 0, top = -1;
  for (int j = 0; j < 16; j++)
    if (low >> j & 1u) bits++, top = j;
  const int rid = (int)((w >> 16) & 31u) + bits - 1;
  if (rid < 0 || rid >= kIdxWords - start0) throw std::runtime_error("index decode: entry without a run");
  const int pos = low ? 16 * r + top : (int)((w >> 21) & 255u);
  return (int)ic[start0 + rid] + (m - pos);
}

// sidx / perm: [ne][P] signed sorted index and tensor-order slot of sorted entry m (make_sub).  Output, padded to a
// multiple of four elements:
//   ic [nep][kIdxWords]      the compressed index (pad elements: one run starting at dof 0)
//   pp [nep][npk + 1][16]    lane t of an element holds entries m = t + 16 r: word k carries the 8-bit slots of
//                            r = 4 k .. 4 k + 3, the last word bit 2 r = flipped, bit 2 r + 1 = only copy
//                            (bits 18 + r: essential, set in the copy stream_set_essential makes)
// Returns false when an element has more than kIdxWords - start0 runs (the caller keeps the one-shot kernel for that block).
inline bool pack_index(int ne, int P, int lsize, const int32_t *sidx, const uint16_t *perm, std::vector<uint32_t> &ic,
                       std::vector<uint32_t> &pp, int start0 = kIdxStart0) {
  if (P > 256) throw std::runtime_error("element too large for 8-bit slots");
  if (lsize >= kExclBit) throw std::runtime_error("too many local dofs for the streaming index encoding");
  const int nep = (ne + 3) & ~3, npl = (P + 15) / 16, npk = (npl + 3) / 4;
  if (npl > 9 || npl > start0) throw std::runtime_error("element too large for the flag word / slice words");
  const int max_runs = kIdxWords - start0;
  const size_t nnz = (size_t)ne * P;
  std::vector<int32_t> count((size_t)lsize, 0);
  for (size_t k = 0; k < nnz; k++) count[dof_of(sidx[k])]++;
  ic.assign((size_t)nep * kIdxWords, 0u);
  pp.assign((size_t)nep * (npk + 1) * 16, 0u);
  for (int e = 0; e < nep; e++) {
    uint32_t *ice = &ic[(size_t)e * kIdxWords];
    if (e >= ne) {
      // one run, dof 0 onwards: the pad entries read the first P entries of x and store to unused E-vector rows
      ice[0] = 1u;
      for (int r = 1; r < npl; r++) ice[r] = 1u << 16;  // one run started before entry 16 r, at position 0
      continue;
    }
    int nruns = 0, lastpos = 0, prev = -2;
    for (int m = 0; m < P; m++) {
      const size_t k = (size_t)e * P + m;
      const int32_t s = sidx[k];
      const int d = dof_of(s);
      const int t = m & 15, r = m >> 4;
      if (t == 0) ice[r] |= (uint32_t)nruns << 16 | (uint32_t)lastpos << 21;
      if (d != prev + 1) {
        if (nruns == max_runs) return false;
        ice[start0 + nruns++] = (uint32_t)d;
        ice[r] |= 1u << t;
        lastpos = m;
      }
      prev = d;
      const bool excl = count[d] == 1;
      uint32_t *row = &pp[(size_t)e * (npk + 1) * 16];
      row[(r >> 2) * 16 + t] |= (uint32_t)(perm[k] & 0xff) << (8 * (r & 3));
      row[npk * 16 + t] |= ((s < 0 ? 1u : 0u) | (excl ? 2u : 0u)) << (2 * r);
    }
  }
  return true;
}

// ---- wide form: one element per 32 lanes (five points per direction, pa_nd_hex_stream5.hip) ----------------------------
// The same idea with 32-entry slices, for elements of up to 320 entries (p = 4: P = 300).  Index block of an element,
// kWideWords = 48 words, fetched by its 32 lanes with two loads and decoded from LDS:
//   word 2 r       entries 32 r .. 32 r + 31: bit j set = entry 32 r + j starts a run
//   word 2 r + 1   bits 0-7 number of runs that start before entry 32 r, bits 8-16 position of the last one
//   word 20 + k    first dof of run k (k < kWideMaxRuns)
// Slot words pp [nep][npk][32], nep = ne padded to two elements (one batch), npk = ceil(npl / 2): lane t of an element
// holds entries m = t + 32 r, word r >> 1 carries the half-words of r = 2 k, 2 k + 1:
//   bits 0-8 tensor-order slot, bit 9 flipped, bit 10 only copy of its dof, bit 11 essential (read as zero; set in the
//   copy stream_set_essential makes, which also takes the entry off the direct path)
constexpr int kWideWords = 48, kWideStart0 = 20, kWideMaxRuns = 24, kWideMaxSlices = 10;
constexpr uint32_t kWideSlotMask = 511u, kWideFlip = 1u << 9, kWideExcl = 1u << 10, kWideEss = 1u << 11;

inline int index_dof_wide(const uint32_t *ic, int m) {  // host model of the device decode
  const int r = m >> 5, t = m & 31;
  const uint32_t w = ic[2 * r], info = ic[2 * r + 1], low = w & ((2u << t) - 1u);
  int bits = 0, top = -1;
  for (int j = 0; j < 32; j++)
    if (low >> j & 1u) bits++, top = j;
  const int rid = (int)(info & 255u) + bits - 1;
  if (rid < 0 || rid >= kWideMaxRuns) throw std::runtime_error("wide index decode: entry without a run");
  const int pos = low ? 32 * r + top : (int)((info >> 8) & 511u);
  return (int)ic[kWideStart0 + rid] + (m - pos);
}

inline bool pack_index_wide(int ne, int P, int lsize, const int32_t *sidx, const uint16_t *perm, std::vector<uint32_t> &ic,
                            std::vector<uint32_t> &pp) {
  if (P > 32 * kWideMaxSlices) throw std::runtime_error("element too large for the wide streaming index");
  if (lsize >= kExclBit) throw std::runtime_error("too many local dofs for the streaming index encoding");
  const int nep = (ne + 1) & ~1, npl = (P + 31) / 32, npk = (npl + 1) / 2;
  const size_t nnz = (size_t)ne * P;
  std::vector<int32_t> count((size_t)lsize, 0);
  for (size_t k = 0; k < nnz; k++) count[dof_of(sidx[k])]++;
  ic.assign((size_t)nep * kWideWords, 0u);
  pp.assign((size_t)nep * npk * 32, 0u);
  for (int e = 0; e < nep; e++) {
    uint32_t *ice = &ic[(size_t)e * kWideWords];
    uint32_t *row = &pp[(size_t)e * npk * 32];
    if (e >= ne) {
      // pad element: one run, dof 0 onwards, every entry read as zero, results stored to E-vector rows nobody gathers
      ice[0] = 1u;
      for (int r = 1; r < npl; r++) ice[2 * r + 1] = 1u;
      for (int m = 0; m < P; m++) row[(m >> 6) * 32 + (m & 31)] |= ((uint32_t)m | kWideEss) << (16 * ((m >> 5) & 1));
      continue;
    }
    int nruns = 0, lastpos = 0, prev = -2;
    for (int m = 0; m < P; m++) {
      const size_t k = (size_t)e * P + m;
      const int32_t s = sidx[k];
      const int d = dof_of(s);
      const int t = m & 31, r = m >> 5;
      if (t == 0) ice[2 * r + 1] = (uint32_t)nruns | (uint32_t)lastpos << 8;
      if (d != prev + 1) {
        if (nruns == kWideMaxRuns) return false;
        ice[kWideStart0 + nruns++] = (uint32_t)d;
        ice[2 * r] |= 1u << t;
        lastpos = m;
      }
      prev = d;
      const uint32_t h = (uint32_t)(perm[k] & kWideSlotMask) | (s < 0 ? kWideFlip : 0u) | (count[d] == 1 ? kWideExcl : 0u);
      row[(r >> 1) * 32 + t] |= h << (16 * (r & 1));
    }
  }
  return true;
}

// Runs over the shared dofs (`shared` increasing: every dof that does not have exactly one copy): consecutive dofs
// with the same number of copies whose copies sit at consecutive E-vector positions, at most 16 long, all essential or none
// (ess: optional flags per dof).
//   hdr[run] = {first dof | length - 1 | essential, first entry in rpos};
//   rpos = E-vector position of the run's first dof in every copy, copies in element order (fixed summation order)
//   code[k] = run << 4 | offset of shared[k] in its run (host-side checks only: the gather kernel walks the headers, sixteen
//   lanes per run)
inline void build_runs(int ne, int P, int lsize, const int32_t *sidx, const std::vector<int32_t> &shared,
                       std::vector<uint32_t> &code, std::vector<RunHdr> &hdr, std::vector<int32_t> &rpos,
                       const char *ess = nullptr, const char *skip = nullptr) {
  // (skip: entries that are not copies -- the pricing experiment of PALACE_AMD_PRICE_BLOCK, pa_nd_hex_stream.hip: build_stream)
  if (lsize > (int)kRunDofMask) throw std::runtime_error("too many local dofs for the run headers");
  const size_t nnz = (size_t)ne * P;
  std::vector<int32_t> tptr((size_t)lsize + 1, 0);
  for (size_t k = 0; k < nnz; k++)
    if (!skip || !skip[k]) tptr[(size_t)dof_of(sidx[k]) + 1]++;
  for (int d = 0; d < lsize; d++) tptr[d + 1] += tptr[d];
  std::vector<int32_t> tpos(nnz), fill(tptr.begin(), tptr.end() - 1);
  for (size_t k = 0; k < nnz; k++)
    if (!skip || !skip[k]) tpos[fill[dof_of(sidx[k])]++] = (int32_t)k;
  code.clear(), hdr.clear(), rpos.clear();
  code.reserve(shared.size());
  int prev = -2, len = 0;
  for (const int32_t d : shared) {
    const int nc = tptr[d + 1] - tptr[d];
    const bool de = ess && ess[d];
    bool extend = (d == prev + 1) && len < 16 && !hdr.empty();
    if (extend) {
      const RunHdr &h = hdr.back();
      extend = (int)rpos.size() - h.ptr == nc && run_ess(h) == de;
      for (int c = 0; extend && c < nc; c++) extend = tpos[tptr[d] + c] == rpos[h.ptr + c] + len;
    }
    if (!extend) {
      hdr.push_back(RunHdr{(uint32_t)d | (de ? 1u << 31 : 0u), (int32_t)rpos.size()});
      for (int c = 0; c < nc; c++) rpos.push_back(tpos[tptr[d] + c]);
      len = 0;
    }
    if (hdr.size() >= (1u << 27)) throw std::runtime_error("too many runs for the gather code");
    code.push_back((uint32_t)(hdr.size() - 1) << 4 | (uint32_t)len);
    hdr.back().dof0 = (hdr.back().dof0 & ~(15u << 27)) | ((uint32_t)len << 27);  // length - 1 so far
    len++, prev = d;
  }
  hdr.push_back(RunHdr{0u, (int32_t)rpos.size()});
}

// Where the shared dofs sit in their runs, 16 bytes per 64 of them: bit i of `starts` = shared dof 64 c + i starts a run (bit 0
// is redundant: `first` says where the chunk's first dof is), first = run of that dof << 4 | its offset in the run.  The gather's
// thread for dof 64 c + i: run = (first >> 4) + popcount(starts bits 1 .. i), offset = i - (highest such bit) or (first & 15) + i.
struct RunChunk {
  unsigned long long starts;
  uint32_t first, pad;
};
inline std::vector<RunChunk> run_chunks(const std::vector<uint32_t> &code) {
  std::vector<RunChunk> ch((code.size() + 63) / 64, RunChunk{0ull, 0u, 0u});
  for (size_t k = 0; k < code.size(); k++) {
    const uint32_t c = code[k] & 0x7fffffffu;
    if ((k & 63) == 0) ch[k >> 6].first = c;
    if ((c & 15u) == 0) ch[k >> 6].starts |= 1ull << (k & 63);
  }
  return ch;
}
// host model of the device decode: {run, offset} of shared dof k
inline void chunk_decode(const std::vector<RunChunk> &ch, size_t k, int &run, int &off) {
  const RunChunk &c = ch[k >> 6];
  const int lane = (int)(k & 63);
  const unsigned long long low = (c.starts & ~1ull) & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  int nc = 0, top = -1;
  for (int i = 0; i < 64; i++)
    if (low >> i & 1ull) nc++, top = i;
  run = (int)(c.first >> 4) + nc;
  off = nc ? lane - top : (int)(c.first & 15u) + lane;
}

// ---- run form of the transpose map of the dense-table path (pa_dense.hip) ------------------------------------------------------
// E-vector position of local dof l of element e: ((e / 16) 4 KP + l) 16 + e % 16 -- consecutive local dofs of an element are 16
// doubles apart.  A run = up to 16 consecutive L-dofs (ALL dofs: the dense gather owns every row) with the same number of copies,
// each copy in the same element with local dofs that step by +1 or -1 (an edge seen against its orientation) and the same sign.
//   hdr[run]  = {first dof | length - 1, first entry in rpos}
//   rpos      = per run and copy (element order: the summation order of the CSR form): position of the run's FIRST dof in that copy
//               | bit 30: the following dofs step backwards | bit 31: the copy enters with a minus sign
//   code[d]   = run << 4 | offset (host checks; the device reads run_chunks(code))
constexpr uint32_t kDenseRunBack = 1u << 30, kDenseRunNeg = 1u << 31, kDenseRunPosMask = kDenseRunBack - 1u;
inline void build_runs_dense(int ne, int P, int KP, int lsize, const int32_t *offsets, const uint8_t *orients,
                             std::vector<uint32_t> &code, std::vector<RunHdr> &hdr, std::vector<uint32_t> &rpos) {
  if (lsize > (int)kRunDofMask) throw std::runtime_error("too many local dofs for the run headers");
  const size_t nnz = (size_t)ne * P;
  std::vector<int32_t> tptr((size_t)lsize + 1, 0);
  for (size_t k = 0; k < nnz; k++) tptr[(size_t)offsets[k] + 1]++;
  for (int d = 0; d < lsize; d++) tptr[d + 1] += tptr[d];
  std::vector<int32_t> tk(nnz), fill(tptr.begin(), tptr.end() - 1);
  for (size_t k = 0; k < nnz; k++) tk[fill[offsets[k]]++] = (int32_t)k;  // copies of a dof in element order
  auto position = [&](int e, int l) { return ((size_t)(e / 16) * 4 * KP + l) * 16 + (size_t)(e % 16); };
  if (position(ne - 1, P - 1) >= kDenseRunBack) throw std::runtime_error("E-vector too large for the dense run positions");
  code.assign((size_t)lsize, 0u), hdr.clear(), rpos.clear();
  std::vector<int32_t> k0;   // first (element, local dof) of every run copy, as e P + l
  std::vector<int8_t> step;  // its local-dof step: 0 not known yet, +1, -1
  int len = 0;
  for (int d = 0; d < lsize; d++) {
    const int nc = tptr[d + 1] - tptr[d];
    bool extend = d > 0 && len > 0 && len < 16 && !hdr.empty();
    if (extend) {
      const RunHdr &h = hdr.back();
      extend = (int)rpos.size() - h.ptr == nc;
      for (int c = 0; extend && c < nc; c++) {
        const int k = tk[tptr[d] + c], e = k / P, l = k % P, e0 = k0[h.ptr + c] / P, l0 = k0[h.ptr + c] % P;
        const bool flip = orients && orients[k], flip0 = (rpos[h.ptr + c] & kDenseRunNeg) != 0;
        const int dl = l - l0;
        extend = e == e0 && flip == flip0 && (step[h.ptr + c] == 0 ? (len == 1 && (dl == 1 || dl == -1)) : dl == step[h.ptr + c] * len);
      }
      if (extend)
        for (int c = 0; c < nc; c++)
          if (step[hdr.back().ptr + c] == 0) step[hdr.back().ptr + c] = (int8_t)(tk[tptr[d] + c] % P - k0[hdr.back().ptr + c] % P);
    }
    if (!extend) {
      hdr.push_back(RunHdr{(uint32_t)d, (int32_t)rpos.size()});
      for (int c = 0; c < nc; c++) {
        const int k = tk[tptr[d] + c];
        k0.push_back(k), step.push_back(0);
        rpos.push_back((uint32_t)position(k / P, k % P) | ((orients && orients[k]) ? kDenseRunNeg : 0u));
      }
      len = 0;
    }
    if (hdr.size() >= (1u << 27)) throw std::runtime_error("too many runs for the gather code");
    code[(size_t)d] = (uint32_t)(hdr.size() - 1) << 4 | (uint32_t)len;
    hdr.back().dof0 = (hdr.back().dof0 & ~(15u << 27)) | ((uint32_t)len << 27);
    len++;
  }
  for (size_t i = 0; i < rpos.size(); i++)
    if (step[i] < 0) rpos[i] |= kDenseRunBack;
  hdr.push_back(RunHdr{0u, (int32_t)rpos.size()});
}

}  // namespace streamhost
}  // namespace pa
