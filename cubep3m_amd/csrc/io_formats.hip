// io_formats.hip -- the reference's particle files (checkpoint.f90, particle_initialization.f90): HOST code only.
// form='unformatted' is Fortran sequential access: every WRITE statement becomes [int32 nbytes][payload][int32 nbytes];
// the reference issues one WRITE per particle.  form='binary' (-DBINARY, an Intel/PGI extension) is the bare payload.
#include "p3m_internal.h"
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
struct File {
  FILE *f = nullptr;
  ~File() { if (f) fclose(f); }
  bool open(const char *path, const char *mode) { f = fopen(path, mode); if (!f) p3m_set_error("cannot open %s", path); return f != nullptr; }
  // writers: an ENOSPC / EIO often only surfaces when stdio's buffer is flushed, so the file counts as written only once
  // both the flush and the close succeeded
  bool finish() { if (!f) return false; const bool ok = fflush(f) == 0; const bool ok2 = fclose(f) == 0; f = nullptr; return ok && ok2; }
};
// one record: payload of n bytes, framed unless binary
bool put(FILE *f, const void *p, size_t n, bool binary) {
  if (!binary && n > 0x7fffffffu) { p3m_set_error("record of %zu bytes does not fit a Fortran sequential record (4-byte markers)", n); return false; }
  const int32_t m = (int32_t)n;
  if (!binary && fwrite(&m, 4, 1, f) != 1) return false;
  if (n && fwrite(p, 1, n, f) != n) return false;
  if (!binary && fwrite(&m, 4, 1, f) != 1) return false;
  return true;
}
bool get(FILE *f, void *p, size_t n, bool binary) {
  int32_t m0 = 0, m1 = 0;
  if (!binary && (fread(&m0, 4, 1, f) != 1 || (size_t)m0 != n)) return false;
  if (n && fread(p, 1, n, f) != n) return false;
  if (!binary && (fread(&m1, 4, 1, f) != 1 || m1 != m0)) return false;
  return true;
}
// the header record: 12 four-byte words with -DPPINT, 11 without (checkpoint.f90:55-61)
size_t pack_header(const p3m_ckpt_header *h, bool ppint, unsigned char *buf) {
  size_t o = 0;
  auto w = [&](const void *v) { memcpy(buf + o, v, 4); o += 4; };
  w(&h->np_local); w(&h->a); w(&h->t); w(&h->tau); w(&h->nts); w(&h->dt_f_acc);
  if (ppint) w(&h->dt_pp_acc);
  w(&h->dt_c_acc); w(&h->cur_checkpoint); w(&h->cur_projection); w(&h->cur_halofind); w(&h->mass_p);
  return o;
}
void unpack_header(const unsigned char *buf, bool ppint, p3m_ckpt_header *h) {
  size_t o = 0;
  auto r = [&](void *v) { memcpy(v, buf + o, 4); o += 4; };
  memset(h, 0, sizeof(*h));
  r(&h->np_local); r(&h->a); r(&h->t); r(&h->tau); r(&h->nts); r(&h->dt_f_acc);
  if (ppint) r(&h->dt_pp_acc);
  r(&h->dt_c_acc); r(&h->cur_checkpoint); r(&h->cur_projection); r(&h->cur_halofind); r(&h->mass_p);
}
int read_header(FILE *f, const char *path, p3m_ckpt_header *h, bool binary, bool ppint) {
  unsigned char buf[48];
  if (!get(f, buf, ppint ? 48 : 44, binary)) { p3m_set_error("%s: bad checkpoint header (wrong layout or -DPPINT setting?)", path); return P3M_EINVAL; }
  unpack_header(buf, ppint, h);
  if (h->np_local < 0) { p3m_set_error("%s: np_local = %d", path, h->np_local); return P3M_EINVAL; }
  return P3M_OK;
}
}  // namespace

extern "C" int p3m_hip_write_checkpoint(const char *path, const p3m_ckpt_header *h, const float *xv6, const float *so, int32_t binary, int32_t ppint) {
  if (!path || !h || (h->np_local > 0 && !xv6)) return P3M_EINVAL;
  File F; if (!F.open(path, "wb")) return P3M_EINVAL;
  unsigned char hb[48];
  bool ok = put(F.f, hb, pack_header(h, ppint != 0, hb), binary != 0);
  const float s0 = so ? so[0] : 0.f, s1 = so ? so[1] : 0.f, s2 = so ? so[2] : 0.f;
  for (int64_t j = 0; ok && j < h->np_local; j++) {          // checkpoint.f90:77-82, one record per particle
    const float *p = xv6 + 6 * j;
    const float rec[6] = {p[0] - s0, p[1] - s1, p[2] - s2, p[3], p[4], p[5]};
    ok = put(F.f, rec, sizeof(rec), binary != 0);
  }
  ok = F.finish() && ok;
  if (!ok) { p3m_set_error("write error on %s", path); return P3M_EINVAL; }
  return P3M_OK;
}
extern "C" int p3m_hip_read_checkpoint(const char *path, p3m_ckpt_header *h, float *xv6, int64_t cap, int32_t binary, int32_t ppint) {
  if (!path || !h) return P3M_EINVAL;
  File F; if (!F.open(path, "rb")) return P3M_EINVAL;
  P3M_TRY(read_header(F.f, path, h, binary != 0, ppint != 0));
  if (!xv6) return P3M_OK;
  if (h->np_local > cap) { p3m_set_error("%s: too many particles to store: np_local %d > %lld (particle_initialization.f90:122-127)", path, h->np_local, (long long)cap); return P3M_ECAPACITY; }
  for (int64_t j = 0; j < h->np_local; j++)
    if (!get(F.f, xv6 + 6 * j, 24, binary != 0)) { p3m_set_error("%s: truncated at particle %lld", path, (long long)j); return P3M_EINVAL; }
  return P3M_OK;
}
extern "C" int p3m_hip_write_pid_checkpoint(const char *path, const p3m_ckpt_header *h, const int64_t *pid, int32_t binary, int32_t ppint) {
  if (!path || !h || (h->np_local > 0 && !pid)) return P3M_EINVAL;
  File F; if (!F.open(path, "wb")) return P3M_EINVAL;
  unsigned char hb[48];
  bool ok = put(F.f, hb, pack_header(h, ppint != 0, hb), binary != 0);
  for (int64_t j = 0; ok && j < h->np_local; j++) ok = put(F.f, pid + j, 8, binary != 0);   // checkpoint.f90:118-123
  ok = F.finish() && ok;
  if (!ok) { p3m_set_error("write error on %s", path); return P3M_EINVAL; }
  return P3M_OK;
}
extern "C" int p3m_hip_read_pid_checkpoint(const char *path, p3m_ckpt_header *h, int64_t *pid, int64_t cap, int32_t binary, int32_t ppint) {
  if (!path || !h) return P3M_EINVAL;
  File F; if (!F.open(path, "rb")) return P3M_EINVAL;
  P3M_TRY(read_header(F.f, path, h, binary != 0, ppint != 0));
  if (!pid) return P3M_OK;
  if (h->np_local > cap) { p3m_set_error("%s: np_local %d > %lld", path, h->np_local, (long long)cap); return P3M_ECAPACITY; }
  for (int64_t j = 0; j < h->np_local; j++)
    if (!get(F.f, pid + j, 8, binary != 0)) { p3m_set_error("%s: truncated at particle %lld", path, (long long)j); return P3M_EINVAL; }
  return P3M_OK;
}
// projection.f90:62-113: `write(u) a` then `write(u) rho_pxy` -- two records (or, -DBINARY, the bare bytes)
extern "C" int p3m_hip_write_projection(const char *path, float a, const float *map, int32_t n, int32_t binary) {
  if (!path || !map || n < 1) return P3M_EINVAL;
  File F; if (!F.open(path, "wb")) return P3M_EINVAL;
  if (!binary && (size_t)4 * n * n > 0x7fffffffu) { p3m_set_error("%s: a %d^2 map does not fit one Fortran sequential record", path, n); return P3M_EINVAL; }
  bool ok = put(F.f, &a, 4, binary != 0) && put(F.f, map, (size_t)4 * n * n, binary != 0);
  ok = F.finish() && ok;
  if (!ok) { p3m_set_error("write error on %s", path); return P3M_EINVAL; }
  return P3M_OK;
}
extern "C" int p3m_hip_read_projection(const char *path, float *a, float *map, int32_t n, int32_t binary) {
  if (!path || !a || !map || n < 1) return P3M_EINVAL;
  File F; if (!F.open(path, "rb")) return P3M_EINVAL;
  if (!get(F.f, a, 4, binary != 0) || !get(F.f, map, (size_t)4 * n * n, binary != 0)) { p3m_set_error("%s: truncated or not a %d^2 projection", path, n); return P3M_EINVAL; }
  return P3M_OK;
}
extern "C" int p3m_hip_write_ic(const char *path, const float *xv6, int32_t np_local, int32_t binary) {
  if (!path || np_local < 0 || (np_local > 0 && !xv6)) return P3M_EINVAL;
  File F; if (!F.open(path, "wb")) return P3M_EINVAL;
  bool ok = put(F.f, &np_local, 4, binary != 0);                                   // dist_init writes np_local first
  if (binary) ok = ok && put(F.f, xv6, (size_t)24 * np_local, true);               // read(20) xv(:,:np_local), :330
  else for (int64_t i = 0; ok && i < np_local; i++) ok = put(F.f, xv6 + 6 * i, 24, false);   // read(20) xv(:,i) per particle, :326-328
  ok = F.finish() && ok;
  if (!ok) { p3m_set_error("write error on %s", path); return P3M_EINVAL; }
  return P3M_OK;
}
extern "C" int p3m_hip_read_ic(const char *path, float *xv6, int64_t cap, int32_t *np_local, int32_t binary) {
  if (!path || !np_local) return P3M_EINVAL;
  File F; if (!F.open(path, "rb")) return P3M_EINVAL;
  int32_t n = 0;
  if (!get(F.f, &n, 4, binary != 0) || n < 0) { p3m_set_error("%s: bad np_local record", path); return P3M_EINVAL; }   // :316
  *np_local = n;
  if (!xv6) return P3M_OK;
  if (n > cap) { p3m_set_error("%s: too many particles to store: np_local %d > max_np %lld (:317-321)", path, n, (long long)cap); return P3M_ECAPACITY; }
  if (binary) { if (!get(F.f, xv6, (size_t)24 * n, true)) { p3m_set_error("%s: truncated", path); return P3M_EINVAL; } }
  else for (int64_t i = 0; i < n; i++) if (!get(F.f, xv6 + 6 * i, 24, false)) { p3m_set_error("%s: truncated at particle %lld", path, (long long)i); return P3M_EINVAL; }
  return P3M_OK;
}

// coarse_power.f90:121-133: <z>ps.dat, formatted, one line '(2f20.10)' per bin: k = 2 pi (bin-1) / box, Delta^2(k)
extern "C" int p3m_hip_write_power(const char *path, const float *ps, int32_t nc_dim) {
  if (!path || !ps || nc_dim < 1) return P3M_EINVAL;
  File F; if (!F.open(path, "w")) return P3M_EINVAL;
  bool ok = true;
  for (int k = 0; ok && k < nc_dim; k++) ok = fprintf(F.f, "%20.10f%20.10f\n", (double)ps[2 * k], (double)ps[2 * k + 1]) > 0;
  ok = F.finish() && ok;
  if (!ok) { p3m_set_error("write error on %s", path); return P3M_EINVAL; }
  return P3M_OK;
}
