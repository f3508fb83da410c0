// CPU check of the streaming kernel's host-side encodings (palace_amd/csrc/pa_stream_host.hpp): a plain C++ model of
// the decode paths of nd_hex_stream_kernel (E staging, E^T stores) and et_run_gather_kernel is run on random signed
// element->dof maps and compared with the definition  y = E^T D E x  (D = a diagonal per-entry scaling).
// Built and run by tests/test_stream_host.py (g++, no GPU).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>

#include "pa_stream_host.hpp"

using namespace pa;
using namespace pa::streamhost;

// min_block: shortest block of consecutive dofs an element is made of (mesh entities in a real numbering); small values
// produce elements with more than kIdxMaxRuns runs, which pack_index has to refuse.
static int run_case(int ne, int P, int lsize, unsigned seed, double ess_frac, int min_block = 6, bool expect_ok = true,
                    int start0 = kIdxStart0) {
  std::mt19937 rng(seed);
  // element -> dof map in tensor order: each element gets P distinct dofs, drawn so that entity-like runs of
  // consecutive dofs are shared between elements (blocks of 1..12 consecutive dofs), random signs
  std::vector<int32_t> lidx((size_t)ne * P);
  for (int e = 0; e < ne; e++) {
    std::vector<char> used(lsize, 0);
    int filled = 0;
    std::vector<int> dofs;
    int tries = 0;
    while (filled < P) {
      // (after many failed placements -- a nearly full dof range -- fall back to single dofs so the loop always ends)
      const int len = tries > 200 ? 1 : std::min<int>(P - filled, min_block + rng() % 12);
      const int d0 = rng() % (lsize - len + 1);
      bool ok = true;
      for (int j = 0; j < len; j++) ok = ok && !used[d0 + j];
      if (!ok) {
        tries++;
        continue;
      }
      for (int j = 0; j < len; j++) used[d0 + j] = 1, dofs.push_back(d0 + j);
      filled += len;
    }
    std::shuffle(dofs.begin(), dofs.end(), rng);
    for (int l = 0; l < P; l++) lidx[(size_t)e * P + l] = (rng() & 1) ? dofs[l] : -1 - dofs[l];
  }
  // sorted order as make_sub builds it
  std::vector<int32_t> sidx((size_t)ne * P);
  std::vector<uint16_t> perm((size_t)ne * P);
  for (int e = 0; e < ne; e++) {
    std::vector<int> ord(P);
    std::iota(ord.begin(), ord.end(), 0);
    const int32_t *le = &lidx[(size_t)e * P];
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return dof_of(le[a]) < dof_of(le[b]); });
    for (int m = 0; m < P; m++) sidx[(size_t)e * P + m] = le[ord[m]], perm[(size_t)e * P + m] = (uint16_t)ord[m];
  }
  std::vector<int32_t> count(lsize, 0);
  for (auto s : sidx) count[dof_of(s)]++;
  std::vector<int32_t> shared;
  for (int d = 0; d < lsize; d++)
    if (count[d] != 1) shared.push_back(d);
  std::vector<char> ess(lsize, 0);
  for (int d = 0; d < lsize; d++) ess[d] = (rng() % 1000) < ess_frac * 1000;

  std::vector<uint32_t> ic, pp;
  const bool ok = pack_index(ne, P, lsize, sidx.data(), perm.data(), ic, pp, start0);
  if (ok != expect_ok) return std::printf("pack_index returned %d, expected %d\n", (int)ok, (int)expect_ok), 1;
  if (!ok) return std::printf("ne=%d P=%d: more than %d runs in an element, refused as expected\n", ne, P, kIdxMaxRuns), 0;
  // the compressed index reproduces every entry's dof
  for (int e = 0; e < ne; e++)
    for (int m = 0; m < P; m++)
      if (index_dof(&ic[(size_t)e * kIdxWords], m, start0) != dof_of(sidx[(size_t)e * P + m]))
        return std::printf("index decode: element %d entry %d: %d != %d\n", e, m, index_dof(&ic[(size_t)e * kIdxWords], m, start0),
                           dof_of(sidx[(size_t)e * P + m])), 1;
  // essential dofs as stream_set_essential handles them: flagged and off the direct path in the flag words, owned by the
  // run list
  const int npl0 = (P + 15) / 16, npk0 = (npl0 + 3) / 4;
  std::vector<uint32_t> ppb(pp);
  for (size_t k = 0; k < (size_t)ne * P; k++) {
    if (ess[dof_of(sidx[k])]) {
      const size_t e = k / P;
      const int m = (int)(k - e * P), t = m & 15, r = m >> 4;
      ppb[(e * (npk0 + 1) + npk0) * 16 + t] &= ~(2u << (2 * r));
      ppb[(e * (npk0 + 1) + npk0) * 16 + t] |= 1u << (18 + r);
    }
  }
  std::vector<int32_t> shared_bc;
  for (int d = 0; d < lsize; d++)
    if (count[d] != 1 || ess[d]) shared_bc.push_back(d);
  std::vector<uint32_t> code;
  std::vector<RunHdr> hdr;
  std::vector<int32_t> rpos;
  build_runs(ne, P, lsize, sidx.data(), shared_bc, code, hdr, rpos, ess.data());
  std::vector<uint32_t> codeb(code);
  for (size_t k = 0; k < codeb.size(); k++)
    if (ess[shared_bc[k]]) codeb[k] |= 0x80000000u;
  {  // what the gather kernel sees: the headers alone name every shared dof exactly once, with its flag
    size_t k = 0;
    for (size_t r = 0; r + 1 < hdr.size(); r++)
      for (int j = 0; j < run_len(hdr[r]); j++, k++)
        if (k >= shared_bc.size() || run_dof0(hdr[r]) + j != shared_bc[k] || run_ess(hdr[r]) != (ess[shared_bc[k]] != 0))
          return std::printf("run headers: run %zu entry %d does not name shared dof %zu\n", r, j, k), 1;
    if (k != shared_bc.size()) return std::printf("run headers: %zu of %zu shared dofs\n", k, shared_bc.size()), 1;
    // ... and the chunk masks place every shared dof in its run (the gather's decode: popcount / count-leading-zeros)
    const std::vector<RunChunk> ch = run_chunks(code);
    for (size_t q = 0; q < code.size(); q++) {
      int run, off;
      chunk_decode(ch, q, run, off);
      if (run != (int)((code[q] & 0x7fffffffu) >> 4) || off != (int)(code[q] & 15u))
        return std::printf("run chunks: shared dof %zu decodes to run %d offset %d, code says %u / %u\n", q, run, off,
                           (code[q] & 0x7fffffffu) >> 4, code[q] & 15u), 1;
    }
  }
  shared = shared_bc;
  pp = ppb;

  std::vector<double> x(lsize), scale((size_t)ne * P);
  std::uniform_real_distribution<double> U(-1, 1);
  for (auto &v : x) v = U(rng);
  for (auto &v : scale) v = U(rng);
  // reference: y = E^T D E (x with essential entries zeroed), then y[ess] = x[ess] (DIAG_ONE)
  std::vector<double> yref(lsize, 0.0);
  for (int e = 0; e < ne; e++)
    for (int l = 0; l < P; l++) {
      const int32_t s = lidx[(size_t)e * P + l];
      const int d = dof_of(s);
      const double u = ess[d] ? 0.0 : (s >= 0 ? x[d] : -x[d]);
      const double v = scale[(size_t)e * P + l] * u;
      yref[d] += s >= 0 ? v : -v;
    }
  for (int d = 0; d < lsize; d++)
    if (ess[d]) yref[d] = x[d];

  // model of the device path
  const int nep = (ne + 3) & ~3, npl = (P + 15) / 16, npk = (npl + 3) / 4;
  std::vector<double> ye((size_t)nep * P, 1e300), y(lsize, 1e300), sm(P);
  for (int e = 0; e < nep; e++) {
    const uint32_t *row = &pp[(size_t)e * (npk + 1) * 16];
    // E: sorted entries into tensor-order slots (kernel: stage loop)
    std::fill(sm.begin(), sm.end(), 0.0);
    for (int t = 0; t < 16; t++)
      for (int r = 0; r < npl; r++) {
        if (t + 16 * r >= P) continue;
        const unsigned fw = row[npk * 16 + t];
        const int dof = index_dof(&ic[(size_t)e * kIdxWords], t + 16 * r, start0);
        if (dof < 0 || dof >= lsize) return std::printf("decoded dof %d out of range\n", dof), 1;
        const double v = (fw >> (18 + r)) & 1u ? 0.0 : x[dof];
        sm[(row[(r >> 2) * 16 + t] >> (8 * (r & 3))) & 255u] = (fw >> (2 * r)) & 1u ? -v : v;
      }
    // element operator: diagonal scaling in tensor order
    if (e < ne)
      for (int l = 0; l < P; l++) sm[l] *= scale[(size_t)e * P + l];
    // E^T stores
    for (int t = 0; t < 16; t++)
      for (int r = 0; r < npl; r++) {
        if (t + 16 * r >= P) continue;
        const unsigned fl = row[npk * 16 + t] >> (2 * r);
        const double v = sm[(row[(r >> 2) * 16 + t] >> (8 * (r & 3))) & 255u];
        const double sgv = (fl & 1u) ? -v : v;
        if (fl & 2u) {
          const int d = index_dof(&ic[(size_t)e * kIdxWords], t + 16 * r, start0);
          if ((row[npk * 16 + t] >> (18 + r)) & 1u) return std::printf("essential dof %d on the direct path\n", d), 1;
          if (e >= ne) return std::printf("pad element %d on the direct path\n", e), 1;
          y[d] = sgv;
        } else {
          ye[(size_t)e * P + t + 16 * r] = sgv;
        }
      }
  }
  for (size_t k = 0; k < shared.size(); k++) {
    const uint32_t c = codeb[k];
    const int run = (int)((c & 0x7fffffffu) >> 4), j = (int)(c & 15u);
    const int d = run_dof0(hdr[run]) + j;
    if (d != shared[k]) return std::printf("run decode: dof %d != %d\n", d, shared[k]), 1;
    if (c >> 31) {
      y[d] = x[d];
      continue;
    }
    double s = 0.0;
    for (int p = hdr[run].ptr; p < hdr[run + 1].ptr; p++) s += ye[(size_t)rpos[p] + j];
    y[d] = s;
  }
  double err = 0.0;
  for (int d = 0; d < lsize; d++) err = std::max(err, std::fabs(y[d] - yref[d]));
  const double avg_run = shared.empty() ? 0.0 : (double)shared.size() / (hdr.size() - 1);
  std::printf("ne=%d P=%d lsize=%d shared=%zu runs=%zu (%.2f dofs/run) max err %.3e\n", ne, P, lsize, shared.size(),
              hdr.size() - 1, avg_run, err);
  return err < 1e-13 ? 0 : 1;
}

// The wide form (pa_nd_hex_stream5.hip: one element per 32 lanes, 16-bit slot half-words with the flags, 48-word index
// block): the same model of E staging, E^T stores and the run gather.
static int run_case_wide(int ne, int P, int lsize, unsigned seed, double ess_frac, int min_block = 6, bool expect_ok = true) {
  std::mt19937 rng(seed);
  std::vector<int32_t> lidx((size_t)ne * P);
  for (int e = 0; e < ne; e++) {
    std::vector<char> used(lsize, 0);
    int filled = 0, tries = 0;
    std::vector<int> dofs;
    while (filled < P) {
      const int len = tries > 200 ? 1 : std::min<int>(P - filled, min_block + rng() % 24);
      const int d0 = rng() % (lsize - len + 1);
      bool ok = true;
      for (int j = 0; j < len; j++) ok = ok && !used[d0 + j];
      if (!ok) {
        tries++;
        continue;
      }
      for (int j = 0; j < len; j++) used[d0 + j] = 1, dofs.push_back(d0 + j);
      filled += len;
    }
    std::shuffle(dofs.begin(), dofs.end(), rng);
    for (int l = 0; l < P; l++) lidx[(size_t)e * P + l] = (rng() & 1) ? dofs[l] : -1 - dofs[l];
  }
  std::vector<int32_t> sidx((size_t)ne * P);
  std::vector<uint16_t> perm((size_t)ne * P);
  for (int e = 0; e < ne; e++) {
    std::vector<int> ord(P);
    std::iota(ord.begin(), ord.end(), 0);
    const int32_t *le = &lidx[(size_t)e * P];
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return dof_of(le[a]) < dof_of(le[b]); });
    for (int m = 0; m < P; m++) sidx[(size_t)e * P + m] = le[ord[m]], perm[(size_t)e * P + m] = (uint16_t)ord[m];
  }
  std::vector<int32_t> count(lsize, 0);
  for (auto s : sidx) count[dof_of(s)]++;
  std::vector<char> ess(lsize, 0);
  for (int d = 0; d < lsize; d++) ess[d] = (rng() % 1000) < ess_frac * 1000;

  std::vector<uint32_t> ic, pp;
  const bool ok = pack_index_wide(ne, P, lsize, sidx.data(), perm.data(), ic, pp);
  if (ok != expect_ok) return std::printf("pack_index_wide returned %d, expected %d\n", (int)ok, (int)expect_ok), 1;
  if (!ok) return std::printf("wide ne=%d P=%d: more than %d runs in an element, refused as expected\n", ne, P, kWideMaxRuns), 0;
  for (int e = 0; e < ne; e++)
    for (int m = 0; m < P; m++)
      if (index_dof_wide(&ic[(size_t)e * kWideWords], m) != dof_of(sidx[(size_t)e * P + m]))
        return std::printf("wide index decode: element %d entry %d\n", e, m), 1;
  const int nep = (ne + 1) & ~1, npl = (P + 31) / 32, npk = (npl + 1) / 2;
  // essential dofs as stream_set_essential handles them in the wide form
  for (size_t k = 0; k < (size_t)ne * P; k++)
    if (ess[dof_of(sidx[k])]) {
      const size_t e = k / P;
      const int m = (int)(k - e * P), t = m & 31, r = m >> 5;
      uint32_t &w = pp[(e * npk + (r >> 1)) * 32 + t];
      w &= ~(kWideExcl << (16 * (r & 1)));
      w |= kWideEss << (16 * (r & 1));
    }
  std::vector<int32_t> shared;
  for (int d = 0; d < lsize; d++)
    if (count[d] != 1 || ess[d]) shared.push_back(d);
  std::vector<uint32_t> code;
  std::vector<RunHdr> hdr;
  std::vector<int32_t> rpos;
  build_runs(ne, P, lsize, sidx.data(), shared, code, hdr, rpos, ess.data());
  for (size_t k = 0; k < code.size(); k++)
    if (ess[shared[k]]) code[k] |= 0x80000000u;

  std::vector<double> x(lsize), scale((size_t)ne * P);
  std::uniform_real_distribution<double> U(-1, 1);
  for (auto &v : x) v = U(rng);
  for (auto &v : scale) v = U(rng);
  std::vector<double> yref(lsize, 0.0);
  for (int e = 0; e < ne; e++)
    for (int l = 0; l < P; l++) {
      const int32_t s = lidx[(size_t)e * P + l];
      const int d = dof_of(s);
      const double u = ess[d] ? 0.0 : (s >= 0 ? x[d] : -x[d]);
      const double v = scale[(size_t)e * P + l] * u;
      yref[d] += s >= 0 ? v : -v;
    }
  for (int d = 0; d < lsize; d++)
    if (ess[d]) yref[d] = x[d];

  std::vector<double> ye((size_t)nep * P, 1e300), y(lsize, 1e300), sm(P);
  auto half = [&](int e, int m) { return (pp[((size_t)e * npk + (m >> 6)) * 32 + (m & 31)] >> (16 * ((m >> 5) & 1))) & 0xffffu; };
  for (int e = 0; e < nep; e++) {
    std::fill(sm.begin(), sm.end(), 0.0);
    for (int m = 0; m < P; m++) {
      const uint32_t h = half(e, m);
      const int dof = index_dof_wide(&ic[(size_t)e * kWideWords], m);
      if (dof < 0 || dof >= lsize) return std::printf("wide: decoded dof %d out of range\n", dof), 1;
      if ((int)(h & kWideSlotMask) >= P) return std::printf("wide: slot out of range\n"), 1;
      const double v = (h & kWideEss) ? 0.0 : x[dof];
      sm[h & kWideSlotMask] = (h & kWideFlip) ? -v : v;
    }
    if (e < ne)
      for (int l = 0; l < P; l++) sm[l] *= scale[(size_t)e * P + l];
    for (int m = 0; m < P; m++) {
      const uint32_t h = half(e, m);
      const double v = sm[h & kWideSlotMask], sgv = (h & kWideFlip) ? -v : v;
      if (h & kWideExcl) {
        if (h & kWideEss) return std::printf("wide: essential dof on the direct path\n"), 1;
        if (e >= ne) return std::printf("wide: pad element %d on the direct path\n", e), 1;
        y[index_dof_wide(&ic[(size_t)e * kWideWords], m)] = sgv;
      } else {
        ye[(size_t)e * P + m] = sgv;
      }
    }
  }
  for (size_t k = 0; k < shared.size(); k++) {
    const uint32_t c = code[k];
    const int run = (int)((c & 0x7fffffffu) >> 4), j = (int)(c & 15u);
    const int d = run_dof0(hdr[run]) + j;
    if (d != shared[k]) return std::printf("wide run decode: dof %d != %d\n", d, shared[k]), 1;
    if (c >> 31) {
      y[d] = x[d];
      continue;
    }
    double s = 0.0;
    for (int p = hdr[run].ptr; p < hdr[run + 1].ptr; p++) s += ye[(size_t)rpos[p] + j];
    y[d] = s;
  }
  double err = 0.0;
  for (int d = 0; d < lsize; d++) err = std::max(err, std::fabs(y[d] - yref[d]));
  std::printf("wide ne=%d P=%d lsize=%d shared=%zu runs=%zu max err %.3e\n", ne, P, lsize, shared.size(), hdr.size() - 1, err);
  return err < 1e-13 ? 0 : 1;
}

// ---- dense path: build_runs_dense against the CSR form of the transpose map -------------------------------------------------------
// Elements of P local dofs made of "entities" of 1 .. 6 consecutive global dofs, each placed at consecutive local positions forwards
// or backwards (an edge seen against its orientation), signs per (element, entity) or -- per_dof_signs -- per entry (runs then break
// inside an entity).  The run form (chunk masks + headers + one position per run and copy) must give exactly the sums of the CSR form.
static int run_case_dense(int ne, int P, int KP, int nent, unsigned seed, bool per_dof_signs) {
  std::mt19937 rng(seed);
  std::vector<int> ent_first, ent_len;
  int lsize = 0;
  for (int k = 0; k < nent; k++) {
    const int len = 1 + (int)(rng() % 6);
    ent_first.push_back(lsize), ent_len.push_back(len), lsize += len;
  }
  lsize += 3;  // a few dofs no element touches (rows the gather writes as zero)
  std::vector<int32_t> offsets((size_t)ne * P);
  std::vector<uint8_t> orients((size_t)ne * P, 0);
  for (int e = 0; e < ne; e++) {
    std::vector<int> order(nent);
    std::iota(order.begin(), order.end(), 0);
    std::shuffle(order.begin(), order.end(), rng);
    int l = 0;
    for (int q = 0; q < nent && l < P; q++) {
      const int k = order[q], len = std::min(ent_len[k], P - l);
      const bool back = rng() & 1u, neg = rng() & 1u;
      for (int i = 0; i < len; i++) {
        offsets[(size_t)e * P + l + i] = ent_first[k] + (back ? len - 1 - i : i);
        orients[(size_t)e * P + l + i] = per_dof_signs ? (uint8_t)(rng() & 1u) : (uint8_t)neg;
      }
      l += len;
    }
    if (l < P) return std::printf("dense case: not enough entities to fill an element\n"), 1;
  }
  const int nb = (ne + 15) / 16;
  std::vector<double> ye((size_t)nb * 4 * KP * 16);
  std::uniform_real_distribution<double> U(-1, 1);
  for (auto &v : ye) v = U(rng);
  auto position = [&](int e, int l) { return ((size_t)(e / 16) * 4 * KP + l) * 16 + (size_t)(e % 16); };
  std::vector<double> yref(lsize, 0.0);
  for (int d = 0; d < lsize; d++) {  // CSR form: copies in element order
    double sum = 0.0;
    for (int e = 0; e < ne; e++)
      for (int l = 0; l < P; l++)
        if (offsets[(size_t)e * P + l] == d) {
          const double v = ye[position(e, l)];
          sum += orients[(size_t)e * P + l] ? -v : v;
        }
    yref[d] = sum;
  }
  std::vector<uint32_t> code, rpos;
  std::vector<RunHdr> hdr;
  build_runs_dense(ne, P, KP, lsize, offsets.data(), orients.data(), code, hdr, rpos);
  const std::vector<RunChunk> ch = run_chunks(code);
  size_t ncopies = 0;
  for (int d = 0; d < lsize; d++) {
    int run, j;
    chunk_decode(ch, (size_t)d, run, j);
    if (run_dof0(hdr[run]) + j != d || j >= run_len(hdr[run])) return std::printf("dense runs: dof %d decodes to run %d offset %d\n", d, run, j), 1;
    double sum = 0.0;
    for (int p = hdr[run].ptr; p < hdr[run + 1].ptr; p++) {
      const uint32_t r = rpos[p];
      const long long at = (long long)(r & kDenseRunPosMask) + ((r & kDenseRunBack) ? -16 * j : 16 * j);
      const double v = ye[(size_t)at];
      sum += (r & kDenseRunNeg) ? -v : v;
    }
    if (sum != yref[d]) return std::printf("dense runs: dof %d: %.17g != %.17g\n", d, sum, yref[d]), 1;
    ncopies += (size_t)(hdr[run + 1].ptr - hdr[run].ptr);
  }
  std::printf("dense ne=%d P=%d lsize=%d runs=%zu positions=%zu (CSR entries %zu) exact\n", ne, P, lsize, hdr.size() - 1, rpos.size(),
              (size_t)ne * P);
  return 0;
}

int main() {
  int bad = 0;
  bad += run_case(37, 144, 2000, 1, 0.05);
  bad += run_case(8, 54, 300, 2, 0.1);
  bad += run_case(5, 12, 40, 3, 0.2);
  bad += run_case(1, 144, 400, 4, 0.0);
  bad += run_case(130, 144, 6000, 5, 0.02);
  bad += run_case(9, 144, 3000, 6, 0.05, 1, false);  // blocks of 1 .. 12 dofs: ~22 runs per element, over the capacity
  bad += run_case(40, 64, 900, 7, 0.05, 1, true, kIdxStart0H1);  // H1 layout: 4 slice words, up to 28 runs
  bad += run_case(11, 27, 300, 8, 0.1, 1, true, kIdxStart0H1);
  bad += run_case_wide(23, 300, 5000, 11, 0.05, 12);  // p = 4
  bad += run_case_wide(7, 144, 1200, 12, 0.1);         // p = 3 on five points
  bad += run_case_wide(6, 54, 400, 13, 0.1);
  bad += run_case_wide(5, 12, 60, 14, 0.2, 1);
  bad += run_case_wide(1, 300, 900, 15, 0.0, 12);
  bad += run_case_wide(4, 300, 6000, 16, 0.05, 1, false);  // short blocks: more than 24 runs, refused
  bad += run_case_dense(40, 45, 12, 60, 21, false);  // order-3 Nedelec tetrahedron: P = 45, KP = 12
  bad += run_case_dense(33, 20, 8, 25, 22, false);
  bad += run_case_dense(19, 45, 12, 40, 23, true);   // signs per entry: runs break inside the entities
  bad += run_case_dense(1, 10, 4, 6, 24, false);
  return bad;
}

// C entry for tests/test_stream_host.py: statistics of the run form on a real element -> dof map
// (lidx: signed tensor-order index [ne][P] as pa_capi.hip builds it).  Returns the number of runs, -1 on error.
extern "C" int stream_host_run_stats(int ne, int P, int lsize, const int32_t *lidx, int *n_shared, int *n_copies) {
  try {
    std::vector<int32_t> sidx((size_t)ne * P);
    for (int e = 0; e < ne; e++) {
      std::vector<int> ord(P);
      std::iota(ord.begin(), ord.end(), 0);
      const int32_t *le = &lidx[(size_t)e * P];
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return dof_of(le[a]) < dof_of(le[b]); });
      for (int m = 0; m < P; m++) sidx[(size_t)e * P + m] = le[ord[m]];
    }
    std::vector<int32_t> count(lsize, 0);
    for (auto s : sidx) count[dof_of(s)]++;
    std::vector<int32_t> shared;
    for (int d = 0; d < lsize; d++)
      if (count[d] != 1) shared.push_back(d);
    std::vector<uint32_t> code;
    std::vector<RunHdr> hdr;
    std::vector<int32_t> rpos;
    build_runs(ne, P, lsize, sidx.data(), shared, code, hdr, rpos);
    *n_shared = (int)shared.size();
    *n_copies = (int)rpos.size();
    return (int)hdr.size() - 1;
  } catch (...) {
    return -1;
  }
}
