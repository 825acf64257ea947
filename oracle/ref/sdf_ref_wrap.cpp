// sdf_ref_wrap.cpp -- C entry point around the REFERENCE's own make_level_set3
// (/root/reference/Tools/SDFGen/makelevelset3.cpp, compiled in place by oracle/Makefile `ref`
// into oracle/_ref/libsdfgen_ref.so; the reference sources are never copied into this repo).
// Test infrastructure only: pins oracle/pa_oracle_sdf.c (and through it the HIP kernels) to the
// reference bit for bit.  Call site replaced: isosurface.cpp:1625-1626.
#include "makelevelset3.h"
#include <cstdint>
#include <vector>

extern "C" int ref_make_level_set3(int64_t ntri, const uint32_t* tri, int64_t nvert, const float* x, const float origin[3], float dx, int ni,
                                   int nj, int nk, float* phi_out, int exact_band) {
  std::vector<Vec3ui> t((size_t)ntri);
  std::vector<Vec3f> v((size_t)nvert);
  for (int64_t q = 0; q < ntri; ++q) t[(size_t)q] = Vec3ui(tri[3 * q], tri[3 * q + 1], tri[3 * q + 2]);
  for (int64_t q = 0; q < nvert; ++q) v[(size_t)q] = Vec3f(x[3 * q], x[3 * q + 1], x[3 * q + 2]);
  Array3f phi;
  make_level_set3(t, v, Vec3f(origin[0], origin[1], origin[2]), dx, ni, nj, nk, phi, exact_band);
  for (int k = 0; k < nk; ++k)
    for (int j = 0; j < nj; ++j)
      for (int i = 0; i < ni; ++i) phi_out[((int64_t)k * nj + j) * ni + i] = phi(i, j, k);
  return 0;
}
