/*
 * p3m_oracle.c -- TEST INFRASTRUCTURE (see p3m_oracle.h): CPU restatement of cubep3m's
 * gravity step.  Paths cited are relative to /root/reference/source_threads (ST/).
 * Arithmetic is fp32 and follows the reference's expression order wherever the reference
 * source fixes it; Fortran 1-based indices are kept in the macros so the code can be read
 * side by side with the .f90 files.
 */
#include "p3m_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float re, im; } cpx;

/* ------------------------------------------------------------------ constants (cubepm.par:148-150) */
static const float PI_F = 3.141592654f;
#define G_F (1.0f / 6.0f / PI_F)
static const float EPS_F = 1.0e-03f;

/* ------------------------------------------------------------------ per-rank state (cubep3m.fh) */
typedef struct {
  float *xv;      /* xv(6,max_np)       cubep3m.fh:75 */
  int64_t *pid;   /* PID(max_np)        cubep3m.fh:77 */
  int *ll;        /* ll(max_np)         cubep3m.fh:76 */
  int *hoc;       /* hoc(hoc_nc_l:hoc_nc_h)^3 cubep3m.fh:78 */
  int np_local;
  int np_buf;     /* deleted by link_list */
  int np_ghost;
  float *rho_c;   /* rho_c(ncn,ncn,ncn) cubep3m.fh:58 */
  float *force_c; /* force_c(3,0:ncn+1,0:ncn+1,0:ncn+1) cubep3m.fh:59 */
  float *send_buf; int64_t *send_pid; int nsend;
  int cart[3];    /* cart_coords(1:3): z,y,x  (mpi_initialization.f90:60-64) */
  int nbr[6];     /* cart_neighbor(1..6) = -z,+z,-y,+y,-x,+x (:66-76) */
  float f_force_max2, pp_force_max, pp_ext_force_max; double f_mesh_mass;
  float c_force_max;
} orc_rank;

struct orc_ctx {
  p3m_params p;
  /* derived, cubepm.par:170-208 */
  int nodes, tiles_node, nc_buf, nc_tile_dim, nc_node_dim, nc_dim, nc_slab;
  int nf_physical_tile_dim, nf_physical_node_dim, hoc_nc_l, hoc_nc_h, hoc_pass_depth, hn;
  int max_np, max_buf;
  orc_rank *r;
  float *kern_f; /* (3,nf/2+1,nf,nf) */
  float *kern_c; /* (3,nc/2+1,nc,nc) = all slabs stacked in z */
  int have_kf, have_kc;
  float dt_f_acc, dt_pp_acc, dt_pp_ext_acc, dt_c_acc;
  double sum_rho_f, sum_rho_c;
};

#define XV(R, c, pp) ((R)->xv[(size_t)((pp) - 1) * 6 + ((c) - 1)])
#define HOC(C, R, i, j, k) \
  ((R)->hoc[((size_t)((k) - (C)->hoc_nc_l) * (C)->hn + ((j) - (C)->hoc_nc_l)) * (C)->hn + ((i) - (C)->hoc_nc_l)])

/* ================================================================== FFT
 * FFTW 2.1.5 (absent) restated: rfftwnd r2c is the unnormalised forward DFT (sign -1) with
 * the half-complex axis first, stored in place in a (n+2,n,n) real array as interleaved
 * (re,im) pairs; c2r is the unnormalised backward DFT (sign +1).  fftw2.f90:19-22,
 * fftw3ds.f90:158-161.  Mixed radix, any length; twiddles computed in double. */
typedef struct { int n; cpx *tw; int nfac; int fac[32]; } fft_plan;

static void plan_init(fft_plan *pl, int n) {
  pl->n = n; pl->tw = (cpx *)malloc(sizeof(cpx) * (size_t)n); pl->nfac = 0;
  for (int k = 0; k < n; k++) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    pl->tw[k].re = (float)cos(a); pl->tw[k].im = (float)sin(a);
  }
  int m = n;
  while (m % 4 == 0) { pl->fac[pl->nfac++] = 4; m /= 4; }
  for (int f = 2; m > 1;) { if (m % f == 0) { pl->fac[pl->nfac++] = f; m /= f; } else f++; }
}
static void plan_free(fft_plan *pl) { free(pl->tw); }

/* recursive decimation in time: out[k + m*q] = sum_j W_n^{jk} W_p^{jm} F_j[k] */
static void fft_rec(const fft_plan *pl, int n, int fi, const cpx *in, int is, cpx *out, int sign) {
  if (n == 1) { out[0] = in[0]; return; }
  int p = pl->fac[fi], q = n / p, N = pl->n, tws = N / n;
  for (int j = 0; j < p; j++) fft_rec(pl, q, fi + 1, in + (size_t)j * is, is * p, out + (size_t)j * q, sign);
  cpx t[32];
  for (int k = 0; k < q; k++) {
    for (int j = 0; j < p; j++) {
      cpx w = pl->tw[((size_t)j * k * tws) % N]; if (sign > 0) w.im = -w.im;
      cpx v = out[(size_t)j * q + k];
      t[j].re = v.re * w.re - v.im * w.im; t[j].im = v.re * w.im + v.im * w.re;
    }
    if (p == 2) {
      out[k].re = t[0].re + t[1].re; out[k].im = t[0].im + t[1].im;
      out[k + q].re = t[0].re - t[1].re; out[k + q].im = t[0].im - t[1].im;
    } else if (p == 4) {
      cpx a = {t[0].re + t[2].re, t[0].im + t[2].im}, b = {t[0].re - t[2].re, t[0].im - t[2].im};
      cpx c = {t[1].re + t[3].re, t[1].im + t[3].im}, d = {t[1].re - t[3].re, t[1].im - t[3].im};
      /* forward: -i*d ; backward: +i*d */
      cpx jd = (sign > 0) ? (cpx){-d.im, d.re} : (cpx){d.im, -d.re};
      out[k] = (cpx){a.re + c.re, a.im + c.im};
      out[k + q] = (cpx){b.re + jd.re, b.im + jd.im};
      out[k + 2 * q] = (cpx){a.re - c.re, a.im - c.im};
      out[k + 3 * q] = (cpx){b.re - jd.re, b.im - jd.im};
    } else {
      int ps = N / p;
      for (int m = 0; m < p; m++) {
        float sr = 0.f, si = 0.f;
        for (int j = 0; j < p; j++) {
          cpx w = pl->tw[((size_t)j * m % p) * ps]; if (sign > 0) w.im = -w.im;
          sr += t[j].re * w.re - t[j].im * w.im; si += t[j].re * w.im + t[j].im * w.re;
        }
        out[k + (size_t)m * q] = (cpx){sr, si};
      }
    }
  }
}
static void fft1d(const fft_plan *pl, const cpx *in, cpx *out, int sign) { fft_rec(pl, pl->n, 0, in, 1, out, sign); }

/* in-place 3-D r2c (dir>0) / c2r incl. no normalisation (dir<0) of a (nx+2,ny,nz) array */
static void fft3d_raw(float *a, int nx, int ny, int nz, int dir) {
  int hx = nx / 2, nxc = hx + 1; size_t pitch = (size_t)nx + 2;
  fft_plan px, py, pz; plan_init(&px, hx); plan_init(&py, ny); plan_init(&pz, nz);
  fft_plan pn; plan_init(&pn, nx); /* only for W_nx^k twiddles */
#pragma omp parallel
  {
    int mx = nx > ny ? nx : ny; if (nz > mx) mx = nz;
    cpx *b0 = (cpx *)malloc(sizeof(cpx) * (size_t)(mx + 2)), *b1 = (cpx *)malloc(sizeof(cpx) * (size_t)(mx + 2));
    if (dir > 0) {
      /* x: real rows via a half-length complex transform */
#pragma omp for collapse(2)
      for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) {
        float *row = a + ((size_t)k * ny + j) * pitch;
        for (int m = 0; m < hx; m++) { b0[m].re = row[2 * m]; b0[m].im = row[2 * m + 1]; }
        fft1d(&px, b0, b1, -1);
        b1[hx] = b1[0];
        for (int m = 0; m <= hx; m++) {
          cpx zk = b1[m], zc = {b1[hx - m].re, -b1[hx - m].im};
          cpx e = {0.5f * (zk.re + zc.re), 0.5f * (zk.im + zc.im)};
          cpx o = {0.5f * (zk.im - zc.im), -0.5f * (zk.re - zc.re)}; /* (zk-zc)/(2i) */
          cpx w = pn.tw[m % nx]; if (m == hx) { w.re = -1.f; w.im = 0.f; }
          row[2 * m] = e.re + (o.re * w.re - o.im * w.im);
          row[2 * m + 1] = e.im + (o.re * w.im + o.im * w.re);
        }
      }
    }
    /* y and z: strided complex lines; order y,z forward and z,y backward */
    for (int pass = 0; pass < 2; pass++) {
      int do_y = (dir > 0) ? (pass == 0) : (pass == 1);
      int sgn = dir > 0 ? -1 : +1;
      if (do_y) {
#pragma omp for collapse(2)
        for (int k = 0; k < nz; k++) for (int i = 0; i < nxc; i++) {
          float *base = a + (size_t)k * ny * pitch + 2 * (size_t)i;
          for (int j = 0; j < ny; j++) { b0[j].re = base[(size_t)j * pitch]; b0[j].im = base[(size_t)j * pitch + 1]; }
          fft1d(&py, b0, b1, sgn);
          for (int j = 0; j < ny; j++) { base[(size_t)j * pitch] = b1[j].re; base[(size_t)j * pitch + 1] = b1[j].im; }
        }
      } else {
        size_t zs = (size_t)ny * pitch;
#pragma omp for collapse(2)
        for (int j = 0; j < ny; j++) for (int i = 0; i < nxc; i++) {
          float *base = a + (size_t)j * pitch + 2 * (size_t)i;
          for (int k = 0; k < nz; k++) { b0[k].re = base[(size_t)k * zs]; b0[k].im = base[(size_t)k * zs + 1]; }
          fft1d(&pz, b0, b1, sgn);
          for (int k = 0; k < nz; k++) { base[(size_t)k * zs] = b1[k].re; base[(size_t)k * zs + 1] = b1[k].im; }
        }
      }
    }
    if (dir < 0) {
#pragma omp for collapse(2)
      for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) {
        float *row = a + ((size_t)k * ny + j) * pitch;
        for (int m = 0; m < hx; m++) {
          cpx xk = {row[2 * m], row[2 * m + 1]}, xc = {row[2 * (hx - m)], -row[2 * (hx - m) + 1]};
          cpx e = {xk.re + xc.re, xk.im + xc.im};
          cpx d = {xk.re - xc.re, xk.im - xc.im};
          cpx w = pn.tw[m]; w.im = -w.im;               /* W_n^{-m} */
          cpx o = {d.re * w.re - d.im * w.im, d.re * w.im + d.im * w.re};
          b0[m].re = e.re - o.im; b0[m].im = e.im + o.re; /* e + i*o */
        }
        fft1d(&px, b0, b1, +1);
        for (int m = 0; m < hx; m++) { row[2 * m] = b1[m].re; row[2 * m + 1] = b1[m].im; }
        row[nx] = 0.f; row[nx + 1] = 0.f;
      }
    }
    free(b0); free(b1);
  }
  plan_free(&px); plan_free(&py); plan_free(&pz); plan_free(&pn);
}

/* cubepm_fftw2 (fftw2.f90:1-31): 'f' forward; 'b' backward then divide by nf^3 */
void orc_fft3d_rect(float *a, int nx, int ny, int nz, int dir) {
  fft3d_raw(a, nx, ny, nz, dir);
  if (dir < 0) {
    float s = (float)nx * (float)ny * (float)nz; /* real(n)**3, fftw2.f90:22 / fftw3ds.f90:161 */
    size_t tot = (size_t)(nx + 2) * ny * nz;
    for (size_t i = 0; i < tot; i++) a[i] = a[i] / s;
  }
}
void orc_fft3d(float *a, int n, int dir) { orc_fft3d_rect(a, n, n, n, dir); }

/* ================================================================== lifecycle */
int64_t orc_derived(const orc_ctx *c, int what) {
  switch (what) {
    case 0: return c->max_np; case 1: return c->nc_dim; case 2: return c->nc_node_dim;
    case 3: return c->nf_physical_node_dim; case 4: return c->nc_slab; case 5: return c->nf_physical_tile_dim;
  }
  return -1;
}

orc_ctx *orc_create(const p3m_params *p) {
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->p = *p;
  int nd = p->nodes_dim, T = p->tiles_node_dim, nf = p->nf_tile, nb = p->nf_buf, ms = p->mesh_scale;
  c->nodes = nd * nd * nd; c->tiles_node = T * T * T;
  c->nc_buf = nb / ms;                                  /* cubepm.par:190 */
  c->nc_tile_dim = (nf - 2 * nb) / ms;                  /* :192 */
  c->nc_node_dim = c->nc_tile_dim * T;                  /* :193 */
  c->nc_dim = c->nc_node_dim * nd;                      /* :194 */
  c->nc_slab = c->nc_dim / c->nodes;                    /* :197 */
  c->nf_physical_tile_dim = nf - 2 * nb;                /* :199 */
  c->nf_physical_node_dim = c->nf_physical_tile_dim * T;/* :202 */
  c->hoc_nc_l = 1 - c->nc_buf; c->hoc_nc_h = c->nc_node_dim + c->nc_buf; /* :205-207 */
  c->hoc_pass_depth = 2 * c->nc_buf;                    /* :208 */
  c->hn = c->hoc_nc_h - c->hoc_nc_l + 1;
  {
    /* cubepm.par:170-172 (integer arithmetic inside, real(4) density_buffer outside) */
    int Nn = c->nf_physical_node_dim;
    double inner = (double)((Nn / 2) * (Nn / 2)) * (double)(Nn / 2) +
                   (8.0 * nb * nb * nb + 6.0 * nb * ((double)Nn * Nn) + 12.0 * ((double)nb * nb) * Nn) / 8.0;
    double mnp = (double)p->density_buffer * inner;
    /* the oracle must also hold every periodic image: never smaller than 2x that count */
    c->max_np = (int)mnp;
    c->max_buf = (int)(2.2 * mnp);                      /* :175 */
  }
  c->r = (orc_rank *)calloc((size_t)c->nodes, sizeof(orc_rank));
  int ncn = c->nc_node_dim;
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
    R->xv = (float *)calloc((size_t)c->max_np * 6, sizeof(float));
    R->pid = (int64_t *)calloc((size_t)c->max_np, sizeof(int64_t));
    R->ll = (int *)calloc((size_t)c->max_np, sizeof(int));
    R->hoc = (int *)calloc((size_t)c->hn * c->hn * c->hn, sizeof(int));
    R->rho_c = (float *)calloc((size_t)ncn * ncn * ncn, sizeof(float));
    R->force_c = (float *)calloc((size_t)3 * (ncn + 2) * (ncn + 2) * (ncn + 2), sizeof(float));
    R->send_buf = (float *)calloc((size_t)c->max_buf, sizeof(float));
    R->send_pid = (int64_t *)calloc((size_t)c->max_buf / 6 + 1, sizeof(int64_t));
    /* MPI_Cart_create, row-major, reorder=.false.: rank = c1*nd^2 + c2*nd + c3 */
    R->cart[0] = rk / (nd * nd); R->cart[1] = (rk / nd) % nd; R->cart[2] = rk % nd;
    for (int d = 0; d < 3; d++) {
      int cm[3] = {R->cart[0], R->cart[1], R->cart[2]}, cp[3] = {R->cart[0], R->cart[1], R->cart[2]};
      cm[d] = (cm[d] - 1 + nd) % nd; cp[d] = (cp[d] + 1) % nd;
      R->nbr[2 * d] = cm[0] * nd * nd + cm[1] * nd + cm[2];     /* cart_neighbor(2d+1): negative */
      R->nbr[2 * d + 1] = cp[0] * nd * nd + cp[1] * nd + cp[2]; /* cart_neighbor(2d+2): positive */
    }
  }
  c->kern_f = (float *)calloc((size_t)3 * (nf / 2 + 1) * nf * nf, sizeof(float));
  c->kern_c = (float *)calloc((size_t)3 * (c->nc_dim / 2 + 1) * c->nc_dim * c->nc_dim, sizeof(float));
  /* variable_initialization.f90:22-29 */
  c->dt_f_acc = c->dt_pp_acc = c->dt_pp_ext_acc = c->dt_c_acc = 1000.f;
  return c;
}

void orc_destroy(orc_ctx *c) {
  if (!c) return;
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
    free(R->xv); free(R->pid); free(R->ll); free(R->hoc); free(R->rho_c); free(R->force_c);
    free(R->send_buf); free(R->send_pid);
  }
  free(c->r); free(c->kern_f); free(c->kern_c); free(c);
}

int orc_set_particles(orc_ctx *c, int rank, const float *xv6, const int64_t *pid, int np) {
  if (np > c->max_np) return P3M_ECAPACITY;
  orc_rank *R = &c->r[rank];
  memcpy(R->xv, xv6, sizeof(float) * 6 * (size_t)np);
  if (pid) memcpy(R->pid, pid, sizeof(int64_t) * (size_t)np);
  else for (int i = 0; i < np; i++) R->pid[i] = i + 1;
  R->np_local = np; return 0;
}
int orc_get_np(orc_ctx *c, int rank) { return c->r[rank].np_local; }
void orc_get_particles(orc_ctx *c, int rank, float *xv6, int64_t *pid) {
  orc_rank *R = &c->r[rank];
  if (xv6) memcpy(xv6, R->xv, sizeof(float) * 6 * (size_t)R->np_local);
  if (pid) memcpy(pid, R->pid, sizeof(int64_t) * (size_t)R->np_local);
}
const float *orc_kern_f(orc_ctx *c) { return c->kern_f; }
const float *orc_kern_c(orc_ctx *c) { return c->kern_c; }
const float *orc_rho_c(orc_ctx *c, int rank) { return c->r[rank].rho_c; }
const float *orc_force_c(orc_ctx *c, int rank) { return c->r[rank].force_c; }

/* ================================================================== kernels
 * fine_kernel, kernel_initialization.f90:2-267.  table16: float[16][16][16][3] in file row
 * order (i fastest: rows are read in loops k,j,i at :25-28). */
void orc_fine_kernel(orc_ctx *c, const float *table16) {
  int nf = c->p.nf_tile, nc = c->p.nf_cutoff, hx = nf / 2 + 1; size_t pitch = (size_t)nf + 2;
  float *rho = (float *)malloc(sizeof(float) * pitch * nf * nf);
#define RF(i, j, k) rho[((size_t)((k) - 1) * nf + ((j) - 1)) * pitch + ((i) - 1)]
  for (int comp = 1; comp <= 3; comp++) {
    memset(rho, 0, sizeof(float) * pitch * nf * nf);                      /* :23 */
    for (int k = 1; k <= nc; k++) for (int j = 1; j <= nc; j++) for (int i = 1; i <= nc; i++)
      RF(i, j, k) = table16[((((size_t)(k - 1) * nc + (j - 1)) * nc + (i - 1)) * 3) + (comp - 1)];
    if ((c->p.flags & P3M_FLAG_PP_EXT)) {                                 /* :38-54 */
      for (int k = 1; k <= c->p.pp_range + 1; k++) for (int j = 1; j <= c->p.pp_range + 1; j++)
        for (int i = 1; i <= c->p.pp_range + 1; i++) RF(i, j, k) = 0.f;
    }
    float sy = (comp == 2) ? -1.f : 1.f, sx = (comp == 1) ? -1.f : 1.f, sz = (comp == 3) ? -1.f : 1.f;
    for (int j = 2; j <= nc; j++)                                         /* :71-73, y */
      for (int k = 1; k <= nc; k++) for (int i = 1; i <= nc; i++) RF(i, nf - j + 2, k) = sy * RF(i, j, k);
    for (int i = 2; i <= nc; i++)                                         /* :77-79, x */
      for (int k = 1; k <= nc; k++) for (int j = 1; j <= nf; j++) RF(nf - i + 2, j, k) = sx * RF(i, j, k);
    for (int k = 2; k <= nc; k++)                                         /* :83-85, z */
      for (int j = 1; j <= nf; j++) for (int i = 1; i <= nf; i++) RF(i, j, nf - k + 2) = sz * RF(i, j, k);
    orc_fft3d(rho, nf, +1);                                               /* :89 */
    for (int k = 1; k <= nf; k++) for (int j = 1; j <= nf; j++) for (int i = 1; i <= hx; i++) /* :93-99 */
      c->kern_f[(((size_t)(k - 1) * nf + (j - 1)) * hx + (i - 1)) * 3 + (comp - 1)] = RF(2 * i, j, k);
  }
#undef RF
  free(rho); c->have_kf = 1;
}

/* coarse_kernel, kernel_initialization.f90:272-732; all ranks' local volumes are assembled
   into the global nc^3 box (x <-> cart_coords(3), :293-298) and the slab FFT is a global FFT. */
static void build_ck_analytic(orc_ctx *c, float *ck /* (3,nc,nc,nc) comp fastest */) {
  int nc = c->nc_dim, ms = c->p.mesh_scale;
  for (int k = 1; k <= nc; k++) {
    float z = (k < nc / 2 + 2) ? (float)(k - 1) : (float)(k - 1 - nc); z = ms * z; /* :304-309 */
    for (int j = 1; j <= nc; j++) {
      float y = (j < nc / 2 + 2) ? (float)(j - 1) : (float)(j - 1 - nc); y = ms * y;
      for (int i = 1; i <= nc; i++) {
        float x = (i < nc / 2 + 2) ? (float)(i - 1) : (float)(i - 1 - nc); x = ms * x;
        float r = sqrtf(x * x + y * y + z * z);
        float *o = &ck[(((size_t)(k - 1) * nc + (j - 1)) * nc + (i - 1)) * 3];
        if (r == 0.0f) { o[0] = o[1] = o[2] = 0.f; }
        else { float r3 = r * r * r; o[0] = -x / r3; o[1] = -y / r3; o[2] = -z / r3; } /* :330-332 */
      }
    }
  }
}

void orc_coarse_kernel(orc_ctx *c, const float *table4 /* [k][j][i][3] rows of wfxyzc.2.ascii */) {
  int nc = c->nc_dim, hx = nc / 2 + 1; size_t pitch = (size_t)nc + 2;
  size_t n3 = (size_t)nc * nc * nc;
  float *ck = (float *)malloc(sizeof(float) * 3 * n3);
  build_ck_analytic(c, ck);
#define CK(cc, i, j, k) ck[((((size_t)((k) - 1)) * nc + ((j) - 1)) * nc + ((i) - 1)) * 3 + ((cc) - 1)]
#define TB(cc, i, j, k) table4[((((size_t)((k) - 1)) * 4 + ((j) - 1)) * 4 + ((i) - 1)) * 3 + ((cc) - 1)]
  /* :366-457.  With nodes_dim==1 the eight octant copies land in one box of size
     nc_node_dim == nc_dim; with nodes_dim>1 each of the eight CORNER ranks writes its octant
     into its local volume, which in global coordinates is again index nc_dim-i+2.  (Ranks
     that are not corners get no table values; for nodes_dim==2 every rank is a corner.) */
  for (int k = 1; k <= 4; k++) for (int j = 1; j <= 4; j++) for (int i = 1; i <= 4; i++)
    for (int cc = 1; cc <= 3; cc++) CK(cc, i, j, k) = TB(cc, i, j, k);
  for (int k = 2; k <= 4; k++) for (int j = 1; j <= 4; j++) for (int i = 1; i <= 4; i++) {
    CK(1, i, j, nc - k + 2) = TB(1, i, j, k); CK(2, i, j, nc - k + 2) = TB(2, i, j, k); CK(3, i, j, nc - k + 2) = -TB(3, i, j, k);
  }
  for (int j = 2; j <= 4; j++) for (int k = 1; k <= 4; k++) for (int i = 1; i <= 4; i++) {
    CK(1, i, nc - j + 2, k) = TB(1, i, j, k); CK(2, i, nc - j + 2, k) = -TB(2, i, j, k); CK(3, i, nc - j + 2, k) = TB(3, i, j, k);
  }
  for (int k = 2; k <= 4; k++) for (int j = 2; j <= 4; j++) for (int i = 1; i <= 4; i++) {
    CK(1, i, nc - j + 2, nc - k + 2) = TB(1, i, j, k); CK(2, i, nc - j + 2, nc - k + 2) = -TB(2, i, j, k); CK(3, i, nc - j + 2, nc - k + 2) = -TB(3, i, j, k);
  }
  for (int i = 2; i <= 4; i++) for (int k = 1; k <= 4; k++) for (int j = 1; j <= 4; j++) {
    CK(1, nc - i + 2, j, k) = -TB(1, i, j, k); CK(2, nc - i + 2, j, k) = TB(2, i, j, k); CK(3, nc - i + 2, j, k) = TB(3, i, j, k);
  }
  for (int k = 2; k <= 4; k++) for (int i = 2; i <= 4; i++) for (int j = 1; j <= 4; j++) {
    CK(1, nc - i + 2, j, nc - k + 2) = -TB(1, i, j, k); CK(2, nc - i + 2, j, nc - k + 2) = TB(2, i, j, k); CK(3, nc - i + 2, j, nc - k + 2) = -TB(3, i, j, k);
  }
  for (int j = 2; j <= 4; j++) for (int i = 2; i <= 4; i++) for (int k = 1; k <= 4; k++) {
    CK(1, nc - i + 2, nc - j + 2, k) = -TB(1, i, j, k); CK(2, nc - i + 2, nc - j + 2, k) = -TB(2, i, j, k); CK(3, nc - i + 2, nc - j + 2, k) = TB(3, i, j, k);
  }
  for (int k = 2; k <= 4; k++) for (int j = 2; j <= 4; j++) for (int i = 2; i <= 4; i++)
    for (int cc = 1; cc <= 3; cc++) CK(cc, nc - i + 2, nc - j + 2, nc - k + 2) = -TB(cc, i, j, k);

  float *slab = (float *)malloc(sizeof(float) * pitch * nc * nc);
  float *tmp = NULL;
#define SL(i, j, k) slab[((size_t)((k) - 1) * nc + ((j) - 1)) * pitch + ((i) - 1)]
  float *cku = NULL;
  if (c->p.flags & P3M_FLAG_LRCKCORR) { /* :465-553: uncorrected kernel transformed into tmp_kern_c */
    cku = (float *)malloc(sizeof(float) * 3 * n3); build_ck_analytic(c, cku);
    tmp = (float *)malloc(sizeof(float) * 3 * pitch * nc * nc);
    for (int cc = 1; cc <= 3; cc++) {
      for (int k = 1; k <= nc; k++) for (int j = 1; j <= nc; j++) for (int i = 1; i <= nc; i++)
        SL(i, j, k) = cku[(((size_t)(k - 1) * nc + (j - 1)) * nc + (i - 1)) * 3 + (cc - 1)];
      orc_fft3d(slab, nc, +1);
      memcpy(tmp + (size_t)(cc - 1) * pitch * nc * nc, slab, sizeof(float) * pitch * nc * nc);
    }
  }
  for (int cc = 1; cc <= 3; cc++) {
    for (int k = 1; k <= nc; k++) for (int j = 1; j <= nc; j++) for (int i = 1; i <= nc; i++) SL(i, j, k) = CK(cc, i, j, k);
    orc_fft3d(slab, nc, +1);                                              /* :560 / :696 */
    if (c->p.flags & P3M_FLAG_LRCKCORR) {                                 /* :562-591 */
      const float *tk = tmp + (size_t)(cc - 1) * pitch * nc * nc;
      for (int k0 = 1; k0 <= nc; k0++) {
        int kz = (k0 < nc / 2 + 2) ? k0 - 1 : k0 - 1 - nc;
        for (int j = 1; j <= nc; j++) {
          int ky = (j < nc / 2 + 2) ? j - 1 : j - 1 - nc;
          for (int i = 1; i <= nc + 2; i += 2) {
            int kx = (i - 1) / 2;
            float kr = sqrtf((float)(kx * kx + ky * ky + kz * kz));
            if (kr <= 8.f) {
              float ka = 2 * sinf(PI_F * kx / (float)nc), kb = 2 * sinf(PI_F * ky / (float)nc), kc = 2 * sinf(PI_F * kz / (float)nc);
              int kk = (cc == 1) ? kx : (cc == 2) ? ky : kz;
              float kq = (cc == 1) ? ka : (cc == 2) ? kb : kc;
              if (kk != 0) {
                float wa = SL(i + 1, j, k0);
                float wb = tk[((size_t)(k0 - 1) * nc + (j - 1)) * pitch + i];
                float wc = 4.f * PI_F * kq / (ka * ka + kb * kb + kc * kc) / 16.f;
                SL(i + 1, j, k0) = wa * (wc / wb);
              }
            }
          }
        }
      }
    }
    for (int k = 1; k <= nc; k++) for (int j = 1; j <= nc; j++) for (int i = 1; i <= hx; i++)    /* :593-599 */
      c->kern_c[(((size_t)(k - 1) * nc + (j - 1)) * hx + (i - 1)) * 3 + (cc - 1)] = SL(2 * i, j, k);
  }
#undef SL
#undef CK
#undef TB
  free(slab); free(ck); free(tmp); free(cku); c->have_kc = 1;
}

/* ================================================================== update_position.f90:68-76 */
void orc_update_position(orc_ctx *c, float dt, float dt_old, const float *offset) {
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
#pragma omp parallel for schedule(static)                                 /* update_position.f90:68 */
    for (int i = 1; i <= R->np_local; i++) for (int d = 1; d <= 3; d++) {
      if (offset) XV(R, d, i) = XV(R, d, i) + XV(R, d + 3, i) * 0.5f * (dt + dt_old) + offset[d - 1]; /* :71 */
      else XV(R, d, i) = XV(R, d, i) + XV(R, d + 3, i) * 0.5f * (dt + dt_old);                       /* :73 */
    }
  }
}

/* ================================================================== link_list.f90:19-53 */
static void link_list_rank(orc_ctx *c, orc_rank *R) {
  float ms = (float)c->p.mesh_scale;
  memset(R->hoc, 0, sizeof(int) * (size_t)c->hn * c->hn * c->hn);
  R->np_buf = 0;
  int pp = 1;
  for (;;) {
    if (pp > R->np_local) break;
    int i = (int)floorf(XV(R, 1, pp) / ms) + 1, j = (int)floorf(XV(R, 2, pp) / ms) + 1, k = (int)floorf(XV(R, 3, pp) / ms) + 1;
    if (i < c->hoc_nc_l || i > c->hoc_nc_h || j < c->hoc_nc_l || j > c->hoc_nc_h || k < c->hoc_nc_l || k > c->hoc_nc_h) {
      memcpy(&XV(R, 1, pp), &XV(R, 1, R->np_local), 6 * sizeof(float));  /* :32 */
      R->pid[pp - 1] = R->pid[R->np_local - 1];
      R->np_local--; R->np_buf++;
      continue;
    }
    R->ll[pp - 1] = HOC(c, R, i, j, k); HOC(c, R, i, j, k) = pp;          /* :48-49 */
    pp++;
  }
}
void orc_link_list(orc_ctx *c) { for (int rk = 0; rk < c->nodes; rk++) link_list_rank(c, &c->r[rk]); }

/* ================================================================== particle_pass.f90 */
/* axis: 1=x,2=y,3=z (index into xv); dirpos: +1 send towards +axis */
static void pass_pack(orc_ctx *c, orc_rank *R, int axis, int dirpos) {
  float rnf_buf = (float)c->p.nf_buf, Nn = (float)c->nf_physical_node_dim;
  int l = c->hoc_nc_l, h = c->hoc_nc_h, d = c->hoc_pass_depth;
  int lo[4], hi[4];
  for (int a = 1; a <= 3; a++) { lo[a] = l; hi[a] = h; }
  if (dirpos > 0) lo[axis] = h - d; else hi[axis] = l + d;               /* :75 / :182 */
  R->nsend = 0;
  for (int k = lo[3]; k <= hi[3]; k++) for (int j = lo[2]; j <= hi[2]; j++) for (int i = lo[1]; i <= hi[1]; i++) {
    int pp = HOC(c, R, i, j, k);
    while (pp != 0) {
      float x = XV(R, axis, pp);
      int sel = dirpos > 0 ? (x >= Nn - rnf_buf) : (x < rnf_buf);        /* :83 / :185 */
      if (sel) {
        memcpy(&R->send_buf[(size_t)R->nsend * 6], &XV(R, 1, pp), 6 * sizeof(float));
        R->send_pid[R->nsend] = R->pid[pp - 1]; R->nsend++;
      }
      pp = R->ll[pp - 1];
    }
  }
}
static int pass_unpack(orc_ctx *c, orc_rank *R, const orc_rank *S, int axis, int dirpos) {
  float rnf_buf = (float)c->p.nf_buf, Nn = (float)c->nf_physical_node_dim;
  if (R->np_local + S->nsend > c->max_np) return P3M_ECAPACITY;          /* :136-139 */
  for (int i = 1; i <= S->nsend; i++) {
    int q = R->np_local + i;
    memcpy(&XV(R, 1, q), &S->send_buf[(size_t)(i - 1) * 6], 6 * sizeof(float));
    R->pid[q - 1] = S->send_pid[i - 1];
    if (dirpos > 0) {
      XV(R, axis, q) = fmaxf(XV(R, axis, q) - Nn, -rnf_buf);             /* :162 */
    } else {
      if (fabsf(XV(R, axis, q)) < EPS_F) XV(R, axis, q) = (XV(R, axis, q) < 0.0f) ? -EPS_F : EPS_F; /* :257-263 */
      XV(R, axis, q) = fminf(XV(R, axis, q) + Nn, Nn + rnf_buf - EPS_F); /* :264-265 */
    }
  }
  R->np_local += S->nsend;
  return 0;
}
static void pass_relink(orc_ctx *c, orc_rank *R, int first) {            /* :274-298 */
  float ms = (float)c->p.mesh_scale;
  int pp = first;
  for (;;) {
    if (pp > R->np_local) break;
    int i = (int)floorf(XV(R, 1, pp) / ms) + 1, j = (int)floorf(XV(R, 2, pp) / ms) + 1, k = (int)floorf(XV(R, 3, pp) / ms) + 1;
    if (i < c->hoc_nc_l || i > c->hoc_nc_h || j < c->hoc_nc_l || j > c->hoc_nc_h || k < c->hoc_nc_l || k > c->hoc_nc_h) {
      memcpy(&XV(R, 1, pp), &XV(R, 1, R->np_local), 6 * sizeof(float)); /* DIAG branch :281-293 */
      R->pid[pp - 1] = R->pid[R->np_local - 1]; R->np_local--; continue;
    }
    R->ll[pp - 1] = HOC(c, R, i, j, k); HOC(c, R, i, j, k) = pp; pp++;
  }
}
int orc_particle_pass(orc_ctx *c) {
  /* order: +x,-x | -y,+y | +z,-z  (particle_pass.f90:69,177,300,406,520,606) */
  static const int order[3][2] = {{+1, -1}, {-1, +1}, {+1, -1}};
  int *np0 = (int *)malloc(sizeof(int) * (size_t)c->nodes);
  for (int rk = 0; rk < c->nodes; rk++) np0[rk] = c->r[rk].np_local;
  /* a rank's two sends of one axis are packed from the SAME hoc (the first direction's
     arrivals are not yet linked), so pack->unpack can be done direction by direction */
  orc_rank *snap = (orc_rank *)malloc(sizeof(orc_rank) * (size_t)c->nodes);
  for (int ax = 1; ax <= 3; ax++) {
    int *first = (int *)malloc(sizeof(int) * (size_t)c->nodes);
    for (int rk = 0; rk < c->nodes; rk++) first[rk] = c->r[rk].np_local + 1;
    for (int s = 0; s < 2; s++) {
      int dirpos = order[ax - 1][s];
      for (int rk = 0; rk < c->nodes; rk++) pass_pack(c, &c->r[rk], ax, dirpos);
      /* copy the send buffers so that a rank being its own neighbour is handled */
      for (int rk = 0; rk < c->nodes; rk++) {
        snap[rk] = c->r[rk];
        snap[rk].send_buf = (float *)malloc(sizeof(float) * 6 * (size_t)(c->r[rk].nsend + 1));
        snap[rk].send_pid = (int64_t *)malloc(sizeof(int64_t) * (size_t)(c->r[rk].nsend + 1));
        memcpy(snap[rk].send_buf, c->r[rk].send_buf, sizeof(float) * 6 * (size_t)c->r[rk].nsend);
        memcpy(snap[rk].send_pid, c->r[rk].send_pid, sizeof(int64_t) * (size_t)c->r[rk].nsend);
      }
      int err = 0;
      for (int rk = 0; rk < c->nodes; rk++) {
        /* x <-> cart dim 3 (nbr[4],nbr[5]); y <-> dim 2 (nbr[2],nbr[3]); z <-> dim 1 (nbr[0],nbr[1]) */
        int d = 3 - ax;
        int src = dirpos > 0 ? c->r[rk].nbr[2 * d] : c->r[rk].nbr[2 * d + 1]; /* receive from the opposite side */
        err |= pass_unpack(c, &c->r[rk], &snap[src], ax, dirpos);
      }
      for (int rk = 0; rk < c->nodes; rk++) { free(snap[rk].send_buf); free(snap[rk].send_pid); }
      if (err) { free(first); free(snap); free(np0); return P3M_ECAPACITY; }
    }
    for (int rk = 0; rk < c->nodes; rk++) pass_relink(c, &c->r[rk], first[rk]);
    free(first);
  }
  for (int rk = 0; rk < c->nodes; rk++) c->r[rk].np_ghost = c->r[rk].np_local - np0[rk];
  free(snap); free(np0);
  return 0;
}

/* ================================================================== fine mesh
 * particle_mesh_threaded.f90:85-628, one tile at a time. */
typedef struct {
  float *rho_f, *cmplx_rho_f, *force_f; /* cubep3m.fh:33-37 (one "thread" slab) */
  int *hoc_fine, *ll_fine; float *pp_ext_force_accum; /* cubep3m.fh:47-50 */
  int *llf; int *ipl; float *pp_force_accum;          /* cubep3m.fh:44-46 */
  int max_llf;
} tile_ws;

static tile_ws *ws_alloc(orc_ctx *c) {
  tile_ws *w = (tile_ws *)calloc(1, sizeof(tile_ws));
  int nf = c->p.nf_tile, pt = c->nf_physical_tile_dim, ms = c->p.mesh_scale; size_t S = (size_t)(nf + 2) * nf * nf;
  w->rho_f = (float *)malloc(sizeof(float) * S); w->cmplx_rho_f = (float *)malloc(sizeof(float) * S);
  w->force_f = (float *)malloc(sizeof(float) * 3 * (size_t)(pt + 3) * (pt + 3) * (pt + 3));
  if (c->p.flags & P3M_FLAG_PP_EXT) {
    int e = pt + 2 * c->p.pp_range;
    w->hoc_fine = (int *)malloc(sizeof(int) * (size_t)e * e * e);
    w->ll_fine = (int *)malloc(sizeof(int) * (size_t)c->max_np);
    w->pp_ext_force_accum = (float *)malloc(sizeof(float) * 3 * (size_t)c->max_np);
  }
  if (c->p.flags & P3M_FLAG_PPINT) {
    w->max_llf = 100000;                                                  /* cubepm.par:181 */
    /* llf(max_llf,4,4,4): grown lazily per bucket instead of 100000*64 ints */
    w->llf = (int *)malloc(sizeof(int) * (size_t)ms * ms * ms * 4096);
    w->ipl = (int *)malloc(sizeof(int) * (size_t)ms * ms * ms);
    w->pp_force_accum = (float *)malloc(sizeof(float) * 3 * 4096);
  }
  return w;
}
static void ws_free(tile_ws *w) {
  free(w->rho_f); free(w->cmplx_rho_f); free(w->force_f); free(w->hoc_fine); free(w->ll_fine);
  free(w->pp_ext_force_accum); free(w->llf); free(w->ipl); free(w->pp_force_accum); free(w);
}

#define RHOF(w, i, j, k) (w)->rho_f[((size_t)((k) - 1) * nf + ((j) - 1)) * (size_t)(nf + 2) + ((i) - 1)]
#define CRHOF(w, i, j, k) (w)->cmplx_rho_f[((size_t)((k) - 1) * nf + ((j) - 1)) * (size_t)(nf + 2) + ((i) - 1)]
/* force_f(3, nb-1:nf-nb+1, ...) */
#define FF(w, cc, i, j, k) (w)->force_f[((((size_t)((k) - fl)) * fn + ((j) - fl)) * fn + ((i) - fl)) * 3 + ((cc) - 1)]

/* deposit of one tile: :100-164 (NGP inline :131-151; CIC fine_cic_mass.f90 / _buffer.f90) */
static void tile_deposit(orc_ctx *c, orc_rank *R, const int tile[3], float mass_p, tile_ws *w) {
  int nf = c->p.nf_tile, nb = c->p.nf_buf, pt = c->nf_physical_tile_dim, nct = c->nc_tile_dim, ncb = c->nc_buf;
  memset(w->rho_f, 0, sizeof(float) * (size_t)(nf + 2) * nf * nf);        /* :100 */
  int cl[3], ch[3]; float offset[3];
  int ngp = (c->p.flags & P3M_FLAG_NGP) != 0;
  for (int d = 0; d < 3; d++) {
    if (ngp) { cl[d] = nct * tile[d] + 2 - ncb; ch[d] = nct * (tile[d] + 1) + ncb - 1; }  /* :120-121 */
    else { cl[d] = nct * tile[d] + 1 - ncb; ch[d] = nct * (tile[d] + 1) + ncb; }          /* :123-124 */
    offset[d] = (float)(-tile[d] * pt + nb);                                              /* :134 */
  }
  for (int k = cl[2]; k <= ch[2]; k++) for (int j = cl[1]; j <= ch[1]; j++) for (int i = cl[0]; i <= ch[0]; i++) {
    int pp = HOC(c, R, i, j, k);
    if (ngp) {
      while (pp != 0) {
        int i1[3];
        for (int d = 0; d < 3; d++) { float x = XV(R, d + 1, pp) + offset[d]; i1[d] = (int)floorf(x) + 1; } /* :139-143 */
        RHOF(w, i1[0], i1[1], i1[2]) = RHOF(w, i1[0], i1[1], i1[2]) + mass_p;                               /* :148 */
        pp = R->ll[pp - 1];
      }
    } else {
      int boundary = (i == cl[0] || i == ch[0] || j == cl[1] || j == ch[1] || k == cl[2] || k == ch[2]);  /* :154-156 */
      while (pp != 0) {
        int i1[3], i2[3]; float dx1[3], dx2[3];
        for (int d = 0; d < 3; d++) {
          float x = XV(R, d + 1, pp) + offset[d];                         /* fine_cic_mass.f90:17-21 */
          i1[d] = (int)floorf(x) + 1; i2[d] = i1[d] + 1; dx1[d] = (float)i1[d] - x; dx2[d] = 1.f - dx1[d];
        }
        dx1[0] = mass_p * dx1[0]; dx2[0] = mass_p * dx2[0];               /* :23-24 */
        for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
          int ii = cx ? i2[0] : i1[0], jj = cy ? i2[1] : i1[1], kk = cz ? i2[2] : i1[2];
          if (boundary && (ii < 1 || ii > nf || jj < 1 || jj > nf || kk < 1 || kk > nf)) continue; /* _buffer.f90:25-53 */
          float wgt = (cx ? dx2[0] : dx1[0]) * (cy ? dx2[1] : dx1[1]) * (cz ? dx2[2] : dx1[2]);
          RHOF(w, ii, jj, kk) = RHOF(w, ii, jj, kk) + wgt;
        }
        pp = R->ll[pp - 1];
      }
    }
  }
}

/* :176-223: forward FFT, three kernel multiplies + inverse FFTs, force extraction, max |F|^2 */
static float tile_force(orc_ctx *c, tile_ws *w) {
  int nf = c->p.nf_tile, nb = c->p.nf_buf, hx = nf / 2 + 1, pt = c->nf_physical_tile_dim;
  int fl = nb - 1, fn = pt + 3; size_t S = (size_t)(nf + 2) * nf * nf;
  orc_fft3d(w->rho_f, nf, +1);                                            /* :176 */
  memcpy(w->cmplx_rho_f, w->rho_f, sizeof(float) * S);                    /* :180 */
  for (int i3 = 1; i3 <= 3; i3++) {
    for (int k = 1; k <= nf; k++) for (int j = 1; j <= nf; j++) for (int i = 1; i <= hx; i++) {
      int ii = 2 * i, im = ii - 1;
      float kf = c->kern_f[(((size_t)(k - 1) * nf + (j - 1)) * hx + (i - 1)) * 3 + (i3 - 1)];
      RHOF(w, im, j, k) = -CRHOF(w, ii, j, k) * kf;                       /* :188 */
      RHOF(w, ii, j, k) = CRHOF(w, im, j, k) * kf;                        /* :189 */
    }
    orc_fft3d(w->rho_f, nf, -1);                                          /* :197 */
    for (int k = fl; k <= nf - nb + 1; k++) for (int j = fl; j <= nf - nb + 1; j++) for (int i = fl; i <= nf - nb + 1; i++)
      FF(w, i3, i, j, k) = RHOF(w, i, j, k);                              /* :202 */
  }
  float fmax2 = 0.f;                                                      /* :208-223 */
  for (int k = fl; k <= nf - nb + 1; k++) for (int j = fl; j <= nf - nb + 1; j++) for (int i = fl; i <= nf - nb + 1; i++) {
    float fm = FF(w, 1, i, j, k) * FF(w, 1, i, j, k) + FF(w, 2, i, j, k) * FF(w, 2, i, j, k) + FF(w, 3, i, j, k) * FF(w, 3, i, j, k);
    if (fm > fmax2) fmax2 = fm;
  }
  return fmax2;
}

/* :227-368: gather + kick (NGP :265 / CIC :289-316) and intra-cell PP (:274-285, :324-361) */
static void tile_velocity(orc_ctx *c, orc_rank *R, const int tile[3], float a_mid, float dt, float mass_p,
                          tile_ws *w, float *pp_force_max) {
  int nf = c->p.nf_tile, nb = c->p.nf_buf, pt = c->nf_physical_tile_dim, nct = c->nc_tile_dim, ms = c->p.mesh_scale;
  int fl = nb - 1, fn = pt + 3; (void)nf;
  int ngp = (c->p.flags & P3M_FLAG_NGP) != 0, ppint = (c->p.flags & P3M_FLAG_PPINT) != 0;
  float offset[3]; for (int d = 0; d < 3; d++) offset[d] = (float)nb - (float)(tile[d] * pt); /* :227 */
  float rsoft = c->p.rsoft, pp_bias = c->p.pp_bias;
  for (int k = tile[2] * nct + 1; k <= (tile[2] + 1) * nct; k++)
    for (int j = tile[1] * nct + 1; j <= (tile[1] + 1) * nct; j++)
      for (int i = tile[0] * nct + 1; i <= (tile[0] + 1) * nct; i++) {
        int pp = HOC(c, R, i, j, k);
        if (ppint) memset(w->ipl, 0, sizeof(int) * (size_t)ms * ms * ms);                        /* :242 */
        while (pp != 0) {
          float x[3]; int i1[3];
          for (int d = 0; d < 3; d++) { x[d] = XV(R, d + 1, pp) + offset[d]; i1[d] = (int)floorf(x[d]) + 1; } /* :248-249 */
          if (ngp) {
            for (int d = 1; d <= 3; d++)                                                         /* :265-266 */
              XV(R, d + 3, pp) = XV(R, d + 3, pp) + FF(w, d, i1[0], i1[1], i1[2]) * a_mid * G_F * dt;
            if (ppint) {                                                                         /* :276-284 */
              int i2[3]; for (int d = 0; d < 3; d++) i2[d] = (i1[d] - 1) % ms + 1;
              int b = ((i2[2] - 1) * ms + (i2[1] - 1)) * ms + (i2[0] - 1);
              w->ipl[b]++;
              if (w->ipl[b] > 4096) { fprintf(stderr, "oracle: exceeded llf bucket\n"); abort(); }
              w->llf[(size_t)b * 4096 + (w->ipl[b] - 1)] = pp;
            }
          } else {
            int i2[3]; float dx1[3], dx2[3];
            for (int d = 0; d < 3; d++) { i2[d] = i1[d] + 1; dx1[d] = (float)i1[d] - x[d]; dx2[d] = 1.0f - dx1[d]; } /* :289-291 */
            for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
              /* order of :293-316: (1,1,1),(2,1,1),(1,2,1),(2,2,1),(1,1,2),... = cx fastest */
              float dVc = a_mid * G_F * dt * (cx ? dx2[0] : dx1[0]) * (cy ? dx2[1] : dx1[1]) * (cz ? dx2[2] : dx1[2]);
              int ii = cx ? i2[0] : i1[0], jj = cy ? i2[1] : i1[1], kk = cz ? i2[2] : i1[2];
              for (int d = 1; d <= 3; d++) XV(R, d + 3, pp) = XV(R, d + 3, pp) + FF(w, d, ii, jj, kk) * dVc;
            }
          }
          pp = R->ll[pp - 1];
        }
        if (ppint && ngp) {                                                                      /* :324-361 */
          for (int b = 0; b < ms * ms * ms; b++) {
            int n = w->ipl[b]; if (n == 0) continue;
            memset(w->pp_force_accum, 0, sizeof(float) * 3 * (size_t)n);                         /* :331 */
            for (int ip = 1; ip <= n - 1; ip++) {
              int pp1 = w->llf[(size_t)b * 4096 + ip - 1];
              for (int jp = ip + 1; jp <= n; jp++) {
                int pp2 = w->llf[(size_t)b * 4096 + jp - 1];
                float sep[3]; for (int d = 0; d < 3; d++) sep[d] = XV(R, d + 1, pp1) - XV(R, d + 1, pp2); /* :336 */
                float rmag = sqrtf(sep[0] * sep[0] + sep[1] * sep[1] + sep[2] * sep[2]);                 /* :339 */
                if (rmag > rsoft) {
                  float rb = rmag * pp_bias, rb3 = rb * rb * rb;
                  for (int d = 0; d < 3; d++) {
                    float force_pp = mass_p * (sep[d] / rb3);                                            /* :344 */
                    w->pp_force_accum[(ip - 1) * 3 + d] -= force_pp; w->pp_force_accum[(jp - 1) * 3 + d] += force_pp;
                    XV(R, d + 4, pp1) = XV(R, d + 4, pp1) - force_pp * a_mid * G_F * dt;               /* :349 */
                    XV(R, d + 4, pp2) = XV(R, d + 4, pp2) + force_pp * a_mid * G_F * dt;               /* :350 */
                  }
                }
              }
            }
            for (int ip = 1; ip <= n; ip++) {                                                    /* :355-358 */
              float *f = &w->pp_force_accum[(ip - 1) * 3];
              float m = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
              if (m > *pp_force_max) *pp_force_max = m;
            }
          }
        }
      }
}

/* :378-624 extended PP on one tile; returns maxval(|pp_ext_force_accum|) over the tile (:617) */
static float tile_pp_ext(orc_ctx *c, orc_rank *R, const int tile[3], float a_mid, float dt, float mass_p, tile_ws *w) {
  int pt = c->nf_physical_tile_dim, ppr = c->p.pp_range, e = pt + 2 * ppr;
  float rsoft = c->p.rsoft, pp_bias = c->p.pp_bias, ncut = (float)c->p.nf_cutoff;
  int fl[3], fh[3];
  for (int d = 0; d < 3; d++) { fl[d] = tile[d] * pt + 1 - ppr; fh[d] = (tile[d] + 1) * pt + ppr; } /* :397-402 */
#define HF(i, j, k) w->hoc_fine[((size_t)((k) - 1) * e + ((j) - 1)) * e + ((i) - 1)]
  memset(w->hoc_fine, 0, sizeof(int) * (size_t)e * e * e);                                          /* :393 */
  for (int pp = 1; pp <= R->np_local; pp++) {                                                       /* :410-438 */
    int i = (int)floorf(XV(R, 1, pp)) + 1, j = (int)floorf(XV(R, 2, pp)) + 1, k = (int)floorf(XV(R, 3, pp)) + 1;
    if (i < fl[0] || i > fh[0] || j < fl[1] || j > fh[1] || k < fl[2] || k > fh[2]) continue;
    w->ll_fine[pp - 1] = HF(i - fl[0] + 1, j - fl[1] + 1, k - fl[2] + 1);
    HF(i - fl[0] + 1, j - fl[1] + 1, k - fl[2] + 1) = pp;
  }
  memset(w->pp_ext_force_accum, 0, sizeof(float) * 3 * (size_t)R->np_local);                        /* :491 */
  if (ppr != 0) {
    for (int k = 1; k <= pt + ppr; k++) for (int j = 1; j <= e; j++) for (int i = 1; i <= e; i++) { /* :496-498 */
      if (HF(i, j, k) == 0) continue;
      for (int kp = k; kp <= k + ppr; kp++) {
        int jp_min = (kp == k) ? j : ((j - ppr <= 0) ? 1 : j - ppr);                                /* :506-511 */
        int jp_max = (j + ppr > e) ? e : j + ppr;
        for (int jp = jp_min; jp <= jp_max; jp++) {
          int ip_min = (kp == k && jp == j) ? i + 1 : ((i - ppr <= 0) ? 1 : i - ppr);               /* :515-520 */
          int ip_max = (i + ppr > e) ? e : i + ppr;
          for (int ip = ip_min; ip <= ip_max; ip++) {
            if (HF(ip, jp, kp) == 0) continue;
            int phys1 = (ppr < i && i <= pt + ppr && ppr < j && j <= pt + ppr && ppr < k && k <= pt + ppr);       /* :576-578 */
            int phys2 = (ppr < ip && ip <= pt + ppr && ppr < jp && jp <= pt + ppr && ppr < kp && kp <= pt + ppr); /* :584-586 */
            for (int pp1 = HF(i, j, k); pp1 != 0; pp1 = w->ll_fine[pp1 - 1])
              for (int pp2 = HF(ip, jp, kp); pp2 != 0; pp2 = w->ll_fine[pp2 - 1]) {
                float sep[3]; for (int d = 0; d < 3; d++) sep[d] = XV(R, d + 1, pp1) - XV(R, d + 1, pp2);         /* :551 */
                float rmag = sqrtf(sep[0] * sep[0] + sep[1] * sep[1] + sep[2] * sep[2]);
                if (rmag > rsoft) {                                                                               /* :558 */
                  float rb = rmag * pp_bias, rb3 = rb * rb * rb;
                  float taper = 1.f;
                  if (!(rmag > ncut + sqrtf(3.0f))) {                                                             /* :559-564 */
                    float q = rb / ncut;
                    taper = 1.f - (7.0f / 4.0f) * (q * q * q) + (3.0f / 4.0f) * (q * q * q * q * q);
                  }
                  for (int d = 0; d < 3; d++) {
                    float force_pp = mass_p * (sep[d] / rb3) * taper;
                    w->pp_ext_force_accum[(size_t)(pp1 - 1) * 3 + d] -= force_pp;                                 /* :571 */
                    w->pp_ext_force_accum[(size_t)(pp2 - 1) * 3 + d] += force_pp;                                 /* :572 */
                    if (phys1) XV(R, d + 4, pp1) = XV(R, d + 4, pp1) - force_pp * a_mid * G_F * dt;             /* :581 */
                    if (phys2) XV(R, d + 4, pp2) = XV(R, d + 4, pp2) + force_pp * a_mid * G_F * dt;             /* :589 */
                  }
                }
              }
          }
        }
      }
    }
  }
#undef HF
  float mx = 0.f;                                                                                   /* :617 */
  for (int pp = 0; pp < R->np_local; pp++) {
    float *f = &w->pp_ext_force_accum[(size_t)pp * 3];
    float m = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    if (m > mx) mx = m;
  }
  return mx;
}

static void tile_coords(const orc_ctx *c, int cur_tile, int tile[3]) {    /* :86-90 */
  int T = c->p.tiles_node_dim;
  tile[2] = (cur_tile - 1) / (T * T);
  int j = cur_tile - tile[2] * T * T;
  tile[1] = (j - 1) / T; j = j - tile[1] * T; tile[0] = j - 1;
}

void orc_fine_mesh(orc_ctx *c, float a_mid, float dt, float mass_p) {
  int nf = c->p.nf_tile, nb = c->p.nf_buf, cores = c->p.cores > 0 ? c->p.cores : 1;
  int ppext = (c->p.flags & P3M_FLAG_PP_EXT) != 0;
  float *tile_ext_max = (float *)calloc((size_t)c->tiles_node, sizeof(float));
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
    float fmax2 = 0.f, ppmax = 0.f; double fmass = 0.0;
    /* the reference's `!$omp do` over tiles (:84-85): per-thread slabs, disjoint velocity writes */
#pragma omp parallel reduction(max : fmax2, ppmax) reduction(+ : fmass)
    {
      tile_ws *w = ws_alloc(c);
#pragma omp for schedule(dynamic, 1)
      for (int cur_tile = 1; cur_tile <= c->tiles_node; cur_tile++) {
        int tile[3]; tile_coords(c, cur_tile, tile);
        tile_deposit(c, R, tile, mass_p, w);
        for (int k = 1 + nb; k <= nf - nb; k++) for (int j = 1 + nb; j <= nf - nb; j++) for (int i = 1 + nb; i <= nf - nb; i++)
          fmass += (double)RHOF(w, i, j, k);                              /* :167-173 */
        float f2 = tile_force(c, w); if (f2 > fmax2) fmax2 = f2;
        float pm = 0.f; tile_velocity(c, R, tile, a_mid, dt, mass_p, w, &pm); if (pm > ppmax) ppmax = pm;
        if (ppext) tile_ext_max[cur_tile - 1] = tile_pp_ext(c, R, tile, a_mid, dt, mass_p, w);
      }
      ws_free(w);
    }
    R->f_force_max2 = fmax2; R->pp_force_max = ppmax; R->f_mesh_mass = fmass;
    /* :617 assigns (does not max) pp_ext_force_max(thread) per tile, so each thread keeps its LAST
       tile; `!$omp do` static schedule: thread t owns a contiguous chunk of tiles */
    float em = 0.f;
    if (ppext) {
      int nt = cores < c->tiles_node ? cores : c->tiles_node, base = c->tiles_node / nt, rem = c->tiles_node % nt, pos = 0;
      for (int t = 0; t < nt; t++) { int len = base + (t < rem ? 1 : 0); pos += len; if (tile_ext_max[pos - 1] > em) em = tile_ext_max[pos - 1]; }
    }
    R->pp_ext_force_max = em;
  }
  free(tile_ext_max);
  /* :643-696 */
  float fm = 0.f, pm = 0.f, em = 0.f; double sm = 0.0;
  for (int rk = 0; rk < c->nodes; rk++) {
    float f = sqrtf(c->r[rk].f_force_max2); if (f > fm) fm = f;
    if (c->r[rk].pp_force_max > pm) pm = c->r[rk].pp_force_max;
    if (c->r[rk].pp_ext_force_max > em) em = c->r[rk].pp_ext_force_max;
    sm += c->r[rk].f_mesh_mass;
  }
  c->dt_f_acc = 1.0f / sqrtf(fmaxf(0.0001f, fm) * a_mid * G_F);                                     /* :652 */
  if (c->p.flags & P3M_FLAG_PPINT) c->dt_pp_acc = sqrtf(c->p.dt_pp_scale * c->p.rsoft) / fmaxf(sqrtf(pm * a_mid * G_F), 1e-3f);     /* :668 */
  if (ppext) c->dt_pp_ext_acc = sqrtf(c->p.dt_pp_scale * c->p.rsoft) / fmaxf(sqrtf(em * a_mid * G_F), 1e-3f);                     /* :692 */
  c->sum_rho_f = sm;
}

/* projection.f90:2-188: CIC fine density of every tile (fine_cic_mass.f90 over the chains of the
   tile's coarse cells plus one on either side, build_projection :147-156), summed along each axis into the global
   nf_physical_dim^2 maps; only the ranks at coordinate 0 of the projected axis contribute (:170-181); the MPI sum over
   ranks (:41-54) adds exact zeros from everyone else.  Maps in the reference's memory order: pxy[y][x], pxz[z][x], pyz[z][y]. */
void orc_projection(orc_ctx *c, float mass_p, float *pxy, float *pxz, float *pyz, double *rho_tot) {
  const int nf = c->p.nf_tile, nb = c->p.nf_buf, pt = c->nf_physical_tile_dim, nct = c->nc_tile_dim, T = c->p.tiles_node_dim;
  const int Nn = c->nf_physical_node_dim, Np = Nn * c->p.nodes_dim;
  memset(pxy, 0, sizeof(float) * (size_t)Np * Np); memset(pxz, 0, sizeof(float) * (size_t)Np * Np); memset(pyz, 0, sizeof(float) * (size_t)Np * Np);
  double tot = 0.0;
  tile_ws *w = ws_alloc(c);
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
    double rho_node = 0.0;
    for (int cur = 1; cur <= T * T * T; cur++) {                            /* :24-32 */
      int tile[3]; tile_coords(c, cur, tile);
      memset(w->rho_f, 0, sizeof(float) * (size_t)(nf + 2) * nf * nf);      /* :139 */
      float offset[3];
      for (int d = 0; d < 3; d++) offset[d] = (float)(-tile[d] * pt + nb);  /* fine_cic_mass.f90:13 */
      for (int k = nct * tile[2]; k <= nct * (tile[2] + 1) + 1; k++)        /* :144-156 */
        for (int j = nct * tile[1]; j <= nct * (tile[1] + 1) + 1; j++)
          for (int i = nct * tile[0]; i <= nct * (tile[0] + 1) + 1; i++) {
            int pp = HOC(c, R, i, j, k);
            while (pp != 0) {
              int i1[3], i2[3]; float dx1[3], dx2[3];
              for (int d = 0; d < 3; d++) {
                float x = XV(R, d + 1, pp) + offset[d];                     /* fine_cic_mass.f90:17-21 */
                i1[d] = (int)floorf(x) + 1; i2[d] = i1[d] + 1; dx1[d] = (float)i1[d] - x; dx2[d] = 1.f - dx1[d];
              }
              dx1[0] = mass_p * dx1[0]; dx2[0] = mass_p * dx2[0];           /* :23-24 */
              for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
                int ii = cx ? i2[0] : i1[0], jj = cy ? i2[1] : i1[1], kk = cz ? i2[2] : i1[2];
                RHOF(w, ii, jj, kk) = RHOF(w, ii, jj, kk) + (cx ? dx2[0] : dx1[0]) * (cy ? dx2[1] : dx1[1]) * (cz ? dx2[2] : dx1[2]);
              }
              pp = R->ll[pp - 1];
            }
          }
      const int os_x = tile[0] * pt + R->cart[2] * Nn, os_y = tile[1] * pt + R->cart[1] * Nn, os_z = tile[2] * pt + R->cart[0] * Nn;   /* :160-165 */
      for (int k = 1; k <= pt; k++) for (int j = 1; j <= pt; j++) for (int i = 1; i <= pt; i++) {   /* :169-186 */
        const float v = RHOF(w, nb + i, nb + j, nb + k);
        if (R->cart[0] == 0) pxy[(size_t)(os_y + j - 1) * Np + (os_x + i - 1)] += v;
        if (R->cart[1] == 0) pxz[(size_t)(os_z + k - 1) * Np + (os_x + i - 1)] += v;
        if (R->cart[2] == 0) pyz[(size_t)(os_z + k - 1) * Np + (os_y + j - 1)] += v;
        rho_node = rho_node + (double)v;
      }
    }
    tot += rho_node;                                                        /* :34-35 */
  }
  ws_free(w);
  if (rho_tot) *rho_tot = tot;
}

/* probes */
void orc_tile_density(orc_ctx *c, int rank, int tx, int ty, int tz, float mass_p, float *rho_f) {
  tile_ws *w = ws_alloc(c); int tile[3] = {tx, ty, tz};
  tile_deposit(c, &c->r[rank], tile, mass_p, w);
  memcpy(rho_f, w->rho_f, sizeof(float) * (size_t)(c->p.nf_tile + 2) * c->p.nf_tile * c->p.nf_tile);
  ws_free(w);
}
/* fine_velocity on a caller-supplied force_f (stands where :202 would have filled it from the FFT): the maximum of
   :208-223 over the force box, then gather + kick + intra-cell PP of one tile (:227-368; twin fine_velocity.f90:39-235).
   out[0] = max |F|^2 over the box, out[1] = pp_force_max of this tile.  Needs link_list (+ particle_pass). */
void orc_tile_velocity(orc_ctx *c, int rank, int tx, int ty, int tz, const float *force_f, float a_mid, float dt, float mass_p, float *out) {
  int pt = c->nf_physical_tile_dim; size_t n = 3 * (size_t)(pt + 3) * (pt + 3) * (pt + 3);
  tile_ws *w = ws_alloc(c); int tile[3] = {tx, ty, tz};
  memcpy(w->force_f, force_f, sizeof(float) * n);
  float fmax2 = 0.f;
  for (size_t i = 0; i < n; i += 3) {
    float fm = w->force_f[i] * w->force_f[i] + w->force_f[i + 1] * w->force_f[i + 1] + w->force_f[i + 2] * w->force_f[i + 2];
    if (fm > fmax2) fmax2 = fm;
  }
  float pm = 0.f;
  tile_velocity(c, &c->r[rank], tile, a_mid, dt, mass_p, w, &pm);
  out[0] = fmax2; out[1] = pm;
  ws_free(w);
}

void orc_tile_force(orc_ctx *c, const float *rho_f, float *force_f, float *force_max2) {
  tile_ws *w = ws_alloc(c); int pt = c->nf_physical_tile_dim;
  memcpy(w->rho_f, rho_f, sizeof(float) * (size_t)(c->p.nf_tile + 2) * c->p.nf_tile * c->p.nf_tile);
  float f2 = tile_force(c, w);
  memcpy(force_f, w->force_f, sizeof(float) * 3 * (size_t)(pt + 3) * (pt + 3) * (pt + 3));
  if (force_max2) *force_max2 = f2;
  ws_free(w);
}

/* ================================================================== coarse mesh */
#define RHOC(R, i, j, k) (R)->rho_c[((size_t)((k) - 1) * ncn + ((j) - 1)) * ncn + ((i) - 1)]
#define FC(R, cc, i, j, k) (R)->force_c[((((size_t)(k)) * (ncn + 2) + (j)) * (ncn + 2) + (i)) * 3 + ((cc) - 1)]

/* coarse_mass.f90:82-99 with coarse_cic_mass.f90:18-69 / coarse_cic_mass_buffer.f90:23-113 */
void orc_coarse_density(orc_ctx *c, float mass_p) {
  int ncn = c->nc_node_dim, ms = c->p.mesh_scale;
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
    memset(R->rho_c, 0, sizeof(float) * (size_t)ncn * ncn * ncn);         /* coarse_mass.f90:23 */
    for (int k0 = 0; k0 <= ms - 1; k0++)                                  /* :82 */
#pragma omp parallel for schedule(dynamic)                                /* :83: planes mesh_scale apart deposit into disjoint cells */
      for (int k = k0; k <= ncn + 1; k += ms) for (int j = 0; j <= ncn + 1; j++) for (int i = 0; i <= ncn + 1; i++) {
        int pp = HOC(c, R, i, j, k);
        int boundary = (i <= 1 || i >= ncn || j <= 1 || j >= ncn || k <= 1 || k >= ncn);  /* :88-90 */
        while (pp != 0) {
          int i1[3], i2[3]; float dx1[3], dx2[3];
          for (int d = 0; d < 3; d++) {
            float x = (1.0f / (float)ms) * XV(R, d + 1, pp) - 0.5f;       /* coarse_cic_mass.f90:18 */
            i1[d] = (int)floorf(x) + 1; i2[d] = i1[d] + 1; dx1[d] = (float)i1[d] - x; dx2[d] = 1.0f - dx1[d];
            if (c->p.flags & P3M_FLAG_COARSE_NGP) { dx1[d] = 0.0f; dx2[d] = 1.0f; }   /* :21-24 (-DCOARSE_NGP) */
          }
          dx1[0] = mass_p * dx1[0]; dx2[0] = mass_p * dx2[0];             /* :32-33 */
          for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
            int ii = cx ? i2[0] : i1[0], jj = cy ? i2[1] : i1[1], kk = cz ? i2[2] : i1[2];
            if (ii < 1 || ii > ncn || jj < 1 || jj > ncn || kk < 1 || kk > ncn) {
              if (boundary) continue;                                     /* _buffer.f90:59-113 clips */
              fprintf(stderr, "oracle: interior coarse CIC out of range\n"); abort();
            }
            RHOC(R, ii, jj, kk) = RHOC(R, ii, jj, kk) + (cx ? dx2[0] : dx1[0]) * (cy ? dx2[1] : dx1[1]) * (cz ? dx2[2] : dx1[2]);
          }
          pp = R->ll[pp - 1];
        }
      }
  }
}

void orc_distribute_force(orc_ctx *c, const float *fg);
/* coarse_force.f90:18-90 + fftw3ds.f90 (cube<->slab is the identity once all ranks live in one
   address space) + coarse_force_buffer.f90:19-63 (periodic halo from the neighbour cubes) */
void orc_coarse_force(orc_ctx *c) {
  int nc = c->nc_dim, ncn = c->nc_node_dim, nd = c->p.nodes_dim, hx = nc / 2 + 1; size_t pitch = (size_t)nc + 2;
  size_t S = pitch * nc * nc;
  float *slab = (float *)malloc(sizeof(float) * S), *cr = (float *)malloc(sizeof(float) * S);
  float *fg = (float *)malloc(sizeof(float) * 3 * (size_t)nc * nc * nc);
#define SL(a, i, j, k) a[((size_t)((k) - 1) * nc + ((j) - 1)) * pitch + ((i) - 1)]
  memset(slab, 0, sizeof(float) * S);
  for (int rk = 0; rk < c->nodes; rk++) {                                 /* pack_slab: x<->cart(3) */
    orc_rank *R = &c->r[rk]; int ox = R->cart[2] * ncn, oy = R->cart[1] * ncn, oz = R->cart[0] * ncn;
    for (int k = 1; k <= ncn; k++) for (int j = 1; j <= ncn; j++) for (int i = 1; i <= ncn; i++) SL(slab, ox + i, oy + j, oz + k) = RHOC(R, i, j, k);
  }
  orc_fft3d(slab, nc, +1);                                                /* coarse_force.f90:18 */
  memcpy(cr, slab, sizeof(float) * S);                                    /* :20 */
  for (int cc = 1; cc <= 3; cc++) {
#pragma omp parallel for schedule(static)                                 /* coarse_force.f90:37,56,75 */
    for (int k = 1; k <= nc; k++) for (int j = 1; j <= nc; j++) for (int i = 1; i <= hx; i++) {
      int ii = 2 * i, im = ii - 1;
      float kc = c->kern_c[(((size_t)(k - 1) * nc + (j - 1)) * hx + (i - 1)) * 3 + (cc - 1)];
      SL(slab, im, j, k) = -SL(cr, ii, j, k) * kc;                        /* :43 */
      SL(slab, ii, j, k) = SL(cr, im, j, k) * kc;                         /* :44 */
    }
    orc_fft3d(slab, nc, -1);                                              /* :50 incl. /nc^3 (fftw3ds.f90:161) */
#pragma omp parallel for schedule(static)
    for (int k = 1; k <= nc; k++) for (int j = 1; j <= nc; j++) for (int i = 1; i <= nc; i++)
      fg[(((size_t)(k - 1) * nc + (j - 1)) * nc + (i - 1)) * 3 + (cc - 1)] = SL(slab, i, j, k);
  }
#undef SL
  (void)nd;
  orc_distribute_force(c, fg);
  free(slab); free(cr); free(fg);
}

/* coarse_power.f90:2-139: the mass power spectrum of the coarse density.  cmplx_rho_c (the
   transform of rho_c kept by coarse_force.f90:20) goes back to real space, becomes the overdensity rho_c/rho_c_mean - 1
   (:27-31), is transformed again (:35), and every mode of the half spectrum adds its power to the bin k1 = ceiling(|k|)
   (w1 = 1, w2 = 0, :92-95), each rank over its own z-slab (:41-47), the ranks' bins summed by mpi_reduce (:109).  Kept as
   written: the sinc^4 deconvolution divides only the IMAGINARY part's square (:96, operator precedence), and the kx = 0
   plane counts each conjugate pair once (:60-61).  ps: (2, nc_dim) as the reference writes it to <z>ps.dat (:112-133):
   ps[2k] = 2 pi (k-1) / box, ps[2k+1] = Delta^2 = 4 pi (k-1)^3 <P>; bins without modes stay 0.  Needs coarse_mass. */
void orc_coarse_power(orc_ctx *c, float mass_p, float box, float *ps) {
  const int nc = c->nc_dim, ncn = c->nc_node_dim, hc = nc / 2, nslab = nc / c->nodes; const size_t pitch = (size_t)nc + 2;
  const size_t S = pitch * nc * nc;
  float *slab = (float *)calloc(S, sizeof(float));
  float *psr = (float *)calloc((size_t)2 * (nc + 2), sizeof(float)), *sum = (float *)calloc((size_t)2 * (nc + 2), sizeof(float));
#define SL(a, i, j, k) a[((size_t)((k) - 1) * nc + ((j) - 1)) * pitch + ((i) - 1)]
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk]; int ox = R->cart[2] * ncn, oy = R->cart[1] * ncn, oz = R->cart[0] * ncn;
    for (int k = 1; k <= ncn; k++) for (int j = 1; j <= ncn; j++) for (int i = 1; i <= ncn; i++) SL(slab, ox + i, oy + j, oz + k) = RHOC(R, i, j, k);
  }
  orc_fft3d(slab, nc, +1);                                                /* cmplx_rho_c, coarse_force.f90:18-20 */
  const float nfp = (float)(c->nf_physical_node_dim * c->p.nodes_dim / 2), fnc = (float)nc;
  const float rho_c_mean = nfp * nfp * nfp * mass_p / (fnc * fnc * fnc);  /* :24 */
  orc_fft3d(slab, nc, -1);                                                /* :29 cubepm_fftw(-1), incl. / nc^3 */
  for (int k = 1; k <= nc; k++) for (int j = 1; j <= nc; j++) for (int i = 1; i <= nc; i++) SL(slab, i, j, k) = SL(slab, i, j, k) / rho_c_mean - 1.0f;   /* :30 */
  for (int k = 1; k <= nc; k++) for (int j = 1; j <= nc; j++) { SL(slab, nc + 1, j, k) = 0.f; SL(slab, nc + 2, j, k) = 0.f; }
  orc_fft3d(slab, nc, +1);                                                /* :35 */
  const float pi = PI_F, n3 = fnc * fnc * fnc;
  for (int rk = 0; rk < c->nodes; rk++) {
    memset(psr, 0, sizeof(float) * 2 * (size_t)(nc + 2));                 /* :40 */
    for (int k = 1; k <= nslab; k++) {
      const int kg = k + nslab * rk;                                      /* :44 */
      const float kz = (kg < hc + 2) ? (float)(kg - 1) : (float)(kg - 1 - nc);
      for (int j = 1; j <= nc; j++) {
        const float ky = (j < hc + 2) ? (float)(j - 1) : (float)(j - 1 - nc);
        for (int i = 1; i <= nc + 2; i += 2) {
          const float kx = (float)(i - 1) / 2.0f;
          const float kr = sqrtf(kx * kx + ky * ky + kz * kz);
          if (kx == 0.f && ky <= 0.f && kz <= 0.f) continue;              /* :60 */
          if (kx == 0.f && ky > 0.f && kz < 0.f) continue;                /* :61 */
          if (kr != 0.0f) {
            const int k1 = (int)ceilf(kr);
            const float x = pi * kx / fnc, y = pi * ky / fnc, z = pi * kz / fnc;
            const float sx = (x == 0.f) ? 1.f : sinf(x) / x, sy = (y == 0.f) ? 1.f : sinf(y) / y, sz = (z == 0.f) ? 1.f : sinf(z) / z;
            const float kernel = sx * sy * sz;
            const float re = SL(slab, i, j, kg) / n3, im = SL(slab, i + 1, j, kg) / n3;
            const float k2 = kernel * kernel, pw = re * re + im * im / (k2 * k2);            /* :96 */
            psr[2 * (k1 - 1)] += 1.0f; psr[2 * (k1 - 1) + 1] += pw;                           /* :97-98, w1 = 1 */
          }
        }
      }
    }
    for (int k = 0; k < 2 * nc; k++) sum[k] += psr[k];                    /* :109 */
  }
  for (int k = 1; k <= nc; k++) {                                         /* :114-119 */
    ps[2 * (k - 1)] = sum[2 * (k - 1)]; ps[2 * (k - 1) + 1] = sum[2 * (k - 1) + 1];
    if (sum[2 * (k - 1)] != 0.f) {
      const float km = (float)k - 1.f;
      ps[2 * (k - 1) + 1] = 4.0f * pi * (km * km * km) * sum[2 * (k - 1) + 1] / sum[2 * (k - 1)];
      ps[2 * (k - 1)] = 2.0f * pi * km / box;
    }
  }
#undef SL
  free(slab); free(psr); free(sum);
}

/* force_c(:,0:ncn+1,...) = own cube (coarse_force.f90:52,71,90) + the 1-cell periodic halo that the six
   mpi_sendrecv_replace calls of coarse_force_buffer.f90:19-63 deliver (x, then y incl. x-halo, then z:
   edges and corners arrive too).  fg: global (3,nc,nc,nc), component fastest. */
void orc_distribute_force(orc_ctx *c, const float *fg) {
  int nc = c->nc_dim, ncn = c->nc_node_dim;
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk]; int ox = R->cart[2] * ncn, oy = R->cart[1] * ncn, oz = R->cart[0] * ncn;
    for (int k = 0; k <= ncn + 1; k++) for (int j = 0; j <= ncn + 1; j++) for (int i = 0; i <= ncn + 1; i++) {
      int gi = ((ox + i - 1) % nc + nc) % nc, gj = ((oy + j - 1) % nc + nc) % nc, gk = ((oz + k - 1) % nc + nc) % nc;
      for (int cc = 1; cc <= 3; cc++) FC(R, cc, i, j, k) = fg[(((size_t)gk * nc + gj) * nc + gi) * 3 + (cc - 1)];
    }
  }
}

/* coarse_max_dt.f90:17-37 and coarse_velocity.f90:137-179 */
void orc_coarse_max_dt_and_velocity(orc_ctx *c, float a_mid, float dt) {
  int ncn = c->nc_node_dim, ms = c->p.mesh_scale;
  float gmax = 0.f;
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk]; float mx = 0.f;
    for (int k = 1; k <= ncn; k++) for (int j = 1; j <= ncn; j++) for (int i = 1; i <= ncn; i++) {
      float f = sqrtf(FC(R, 1, i, j, k) * FC(R, 1, i, j, k) + FC(R, 2, i, j, k) * FC(R, 2, i, j, k) + FC(R, 3, i, j, k) * FC(R, 3, i, j, k));
      if (f > mx) mx = f;
    }
    R->c_force_max = mx; if (mx > gmax) gmax = mx;
  }
  c->dt_c_acc = sqrtf((float)ms / (gmax * a_mid * G_F));                  /* coarse_max_dt.f90:36 */
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
#pragma omp parallel for schedule(dynamic)                                /* coarse_velocity.f90:137: every particle is in one chain */
    for (int k = 1; k <= ncn; k++) for (int j = 1; j <= ncn; j++) for (int i = 1; i <= ncn; i++) {
      int pp = HOC(c, R, i, j, k);
      while (pp != 0) {
        int i1[3], i2[3]; float dx1[3], dx2[3];
        for (int d = 0; d < 3; d++) {
          float x = (1.0f / (float)ms) * XV(R, d + 1, pp) - 0.5f;         /* coarse_velocity.f90:143 */
          i1[d] = (int)floorf(x) + 1; i2[d] = i1[d] + 1; dx1[d] = (float)i1[d] - x; dx2[d] = 1.0f - dx1[d];
          if (c->p.flags & P3M_FLAG_COARSE_NGP) { dx1[d] = 0.0f; dx2[d] = 1.0f; }     /* :146-149 (-DCOARSE_NGP) */
        }
        for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
          float dV = a_mid * G_F * dt * (cx ? dx2[0] : dx1[0]) * (cy ? dx2[1] : dx1[1]) * (cz ? dx2[2] : dx1[2]); /* :153-167 */
          int ii = cx ? i2[0] : i1[0], jj = cy ? i2[1] : i1[1], kk = cz ? i2[2] : i1[2];
          for (int d = 1; d <= 3; d++) XV(R, d + 3, pp) = XV(R, d + 3, pp) + FC(R, d, ii, jj, kk) * dV;
        }
        pp = R->ll[pp - 1];
      }
    }
  }
}

void orc_coarse_mesh(orc_ctx *c, float a_mid, float dt, float mass_p) {   /* coarse_mesh.f90:29-106 */
  int ncn = c->nc_node_dim;
  orc_coarse_density(c, mass_p);
  double s = 0.0;                                                         /* :31-43 */
  for (int rk = 0; rk < c->nodes; rk++) for (size_t i = 0; i < (size_t)ncn * ncn * ncn; i++) s += (double)c->r[rk].rho_c[i];
  c->sum_rho_c = s;
  orc_coarse_force(c);
  orc_coarse_max_dt_and_velocity(c, a_mid, dt);
}

/* move_grid_back.f90:17-24 */
void orc_move_grid_back(orc_ctx *c, const float *shake_offset) {
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
    for (int i = 1; i <= R->np_local; i++) for (int d = 1; d <= 3; d++) XV(R, d, i) = XV(R, d, i) - shake_offset[d - 1];
  }
}

/* delete_particles.f90:17-47 */
void orc_delete_particles(orc_ctx *c) {
  float Nn = (float)c->nf_physical_node_dim;
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk];
    int pp = 1;
    for (;;) {
      if (pp > R->np_local) break;
      if (XV(R, 1, pp) >= Nn || XV(R, 1, pp) < 0.0f || XV(R, 2, pp) >= Nn || XV(R, 2, pp) < 0.0f ||
          XV(R, 3, pp) >= Nn || XV(R, 3, pp) < 0.0f) {
        memcpy(&XV(R, 1, pp), &XV(R, 1, R->np_local), 6 * sizeof(float));
        R->pid[pp - 1] = R->pid[R->np_local - 1];
        R->np_local--; continue;
      }
      pp++;
    }
  }
}

void orc_step_out(orc_ctx *c, float a_mid, p3m_step_out *out) {
  (void)a_mid;
  memset(out, 0, sizeof(*out));
  out->dt_f_acc = c->dt_f_acc; out->dt_pp_acc = c->dt_pp_acc; out->dt_pp_ext_acc = c->dt_pp_ext_acc; out->dt_c_acc = c->dt_c_acc;
  out->sum_rho_f = c->sum_rho_f; out->sum_rho_c = c->sum_rho_c;
  int64_t tot = 0; float fm = 0.f, pm = 0.f, em = 0.f, cm = 0.f; int ng = 0, ndel = 0;
  for (int rk = 0; rk < c->nodes; rk++) {
    orc_rank *R = &c->r[rk]; tot += R->np_local; ng += R->np_ghost; ndel += R->np_buf;
    float f = sqrtf(R->f_force_max2); if (f > fm) fm = f;
    if (R->pp_force_max > pm) pm = R->pp_force_max;
    if (R->pp_ext_force_max > em) em = R->pp_ext_force_max;
    if (R->c_force_max > cm) cm = R->c_force_max;
  }
  out->np_total = tot; out->np_local = c->r[0].np_local; out->np_ghost = ng; out->np_deleted = ndel;
  out->f_force_max = fm; out->pp_force_max = pm; out->pp_ext_force_max = em; out->c_force_max = cm;
}

int orc_particle_mesh(orc_ctx *c, float a_mid, float dt, float dt_old, float mass_p,
                      const float *offset, const float *move_back, p3m_step_out *out) {
  if (!c->have_kf || !c->have_kc) return P3M_ESTATE;
  orc_update_position(c, dt, dt_old, offset);                             /* :56 */
  orc_link_list(c);                                                       /* :61 */
  int e = orc_particle_pass(c); if (e) return e;                          /* :63 */
  orc_fine_mesh(c, a_mid, dt, mass_p);                                    /* :72-696 */
  orc_coarse_mesh(c, a_mid, dt, mass_p);                                  /* :712 */
  if ((c->p.flags & P3M_FLAG_MOVE_GRID_BACK) && move_back) orc_move_grid_back(c, move_back); /* :716 */
  int ng = 0; for (int rk = 0; rk < c->nodes; rk++) ng += c->r[rk].np_ghost;
  orc_delete_particles(c);                                                /* :720 */
  if (out) orc_step_out(c, a_mid, out);
  return 0;
}

/* ===================================================================== host time loop (timestep.f90)
 * TEST INFRASTRUCTURE like everything in this file.  Pinned bit for bit against the reference's own
 * timestep.o (oracle/_ref, tests/golden/ref_timestep.npz). */
static float orc_half_expansion(const p3m_time_params *P, double ax, float dtx) {
  /* timestep.f90:237-253 (and :255-271 with a_x = a0 + da1); real(8) a_x, adot, addot, atdot, arkm, a3rlm, omHsq */
  double omHsq = (double)(4.0f / 9.0f);
  double a3rlm = pow(ax, (double)(-(3 * P->wde)));
  a3rlm = a3rlm * P->omega_l;
  a3rlm = a3rlm / P->omega_m;
  double arkm = ax * ((1.0f - P->omega_m) - P->omega_l);
  arkm = arkm / P->omega_m;
  double adot = sqrt((omHsq * (ax * ax * ax)) * ((1.0 + arkm) + a3rlm));
  float c2 = 1.5f * (1.0f - P->wde);
  float c3 = (1.5f * (2.0f - 3.0f * P->wde)) * (1.0f - P->wde);
  double addot = ((ax * ax) * omHsq) * ((1.5 + 2.0 * arkm) + c2 * a3rlm);
  double atdot = ((ax * adot) * omHsq) * ((3.0 + 6.0 * arkm) + c3 * a3rlm);
  float d2 = dtx * dtx;
  float d3 = d2 * dtx;
  double s = adot * dtx;
  s = s + (addot * d2) / 2.0;
  s = s + (atdot * d3) / 6.0;
  return (float)s;
}
/* subroutine Chaplygin (timestep.f90:296-339), statement by statement; a3rchm, arkm, G_ch keep their a_x = a0 values in the
 * second half step, as the reference's text has it (:321-329 only assigns a_x) */
static float orc_chap_half(const p3m_time_params *P, double ax, float dtx, double a3rchm, double arkm, double G_ch) {
  double omHsq = (double)(4.0f / 9.0f);
  float one_al = 1.0f + P->alpha_ch;
  double gp = pow(G_ch, (double)(1.0f / one_al));
  double adot = sqrt((omHsq * (ax * ax * ax)) * ((1.0 + arkm) + a3rchm * gp));                                   /* :314 */
  double gq = pow(G_ch, (double)(-P->alpha_ch / one_al));
  double addot = ((ax * ax) * omHsq) * ((1.5 + 2.0 * arkm) + ((3.0 * a3rchm) * P->A_ch) * gq);                   /* :315 */
  double gr = pow(G_ch, (double)(1.0f / one_al - 2.0f));
  float e1 = -3.0f - 3.0f * P->alpha_ch, e2 = -6.0f - 6.0f * P->alpha_ch;
  float c5 = 5.0f * (P->A_ch * P->A_ch), c3 = (3.0f * P->A_ch) * (1.0f - P->A_ch), ch = 2.0f + P->alpha_ch / 2.0f;
  float c1 = (1.0f - P->A_ch) * (1.0f - P->A_ch);
  double poly = (c5 + (c3 * pow(ax, (double)e1)) * ch) + c1 * pow(ax, (double)e2);
  double atdot = ((ax * adot) * omHsq) * ((3.0 + 6.0 * arkm) + ((3.0 * a3rchm) * gr) * poly);                    /* :316 */
  float d2 = dtx * dtx, d3 = d2 * dtx;
  double s = adot * dtx;
  s = s + (addot * d2) / 2.0;
  s = s + (atdot * d3) / 6.0;                                                                                    /* :319 */
  return (float)s;
}
void orc_expansion(const p3m_time_params *P, float a0, float dt0, float *da1, float *da2) {
  if (P->chaplygin) {                                                                                            /* :251-252 */
    float dtx = dt0 / 2;
    double ax = a0;
    double a3rchm = pow(ax, -3.0) * P->omega_ch;                                                                 /* :310 */
    a3rchm = a3rchm / P->omega_m;
    double arkm = ax * ((1.0f - P->omega_m) - P->omega_ch);                                                      /* :311 */
    arkm = arkm / P->omega_m;
    double G_ch = P->A_ch + (1.0f - P->A_ch) * pow(ax, (double)(-3.0f - 3.0f * P->alpha_ch));                    /* :312 */
    *da1 = orc_chap_half(P, ax, dtx, a3rchm, arkm, G_ch);
    float a1 = a0 + *da1;                                                                                        /* :321 */
    *da2 = orc_chap_half(P, a1, dtx, a3rchm, arkm, G_ch);
    return;
  }
  float dtx = dt0 / 2;
  *da1 = orc_half_expansion(P, a0, dtx);
  float a1 = a0 + *da1;
  *da2 = orc_half_expansion(P, a1, dtx);
}
void orc_timestep(const p3m_time_params *P, unsigned flags, p3m_time_state *S, float dt_f_acc, float dt_pp_acc, float dt_pp_ext_acc,
                  float dt_c_acc) {
  float da_1 = 0.f, da_2 = 0.f, ra, dt_e, am, dt;
  int n;
  S->nts += 1;                                                     /* :20 */
  if (S->nts != 1) S->dt_old = S->dt;                              /* :21 */
  if (!P->cosmo) {                                                 /* :197-216 */
    S->a = 1.0f; S->a_mid = S->a; S->da = 0.0f;
    if ((flags & P3M_FLAG_PPINT) && P->pair_infall) {              /* :204-206 */
      dt = 0.05f / sqrtf(G_F * P->mass_p / (P->cur_sep * P->cur_sep));
      if (dt_f_acc < dt) dt = dt_f_acc;
      if (dt_pp_acc < dt) dt = dt_pp_acc;
      if (dt_c_acc < dt) dt = dt_c_acc;
    } else {                                                       /* :208-214 */
      dt = 1.0f;
      if (dt_f_acc < dt) dt = dt_f_acc;
      if ((flags & P3M_FLAG_PPINT) && dt_pp_acc < dt) dt = dt_pp_acc;
      if ((flags & P3M_FLAG_PPINT) && (flags & P3M_FLAG_PP_EXT) && dt_pp_ext_acc < dt) dt = dt_pp_ext_acc;
      if (dt_c_acc < dt) dt = dt_c_acc;
    }
    if (P->pairwise_ic) dt = 1.0f;                                 /* :210 */
    if (P->shake_test_ic) dt = 1.0f;                               /* :211 */
    S->dt = dt; S->t += dt;                                        /* :212 */
    return;
  }
  dt_e = P->dt_max;                                                /* :59 */
  n = 0;
  for (;;) {                                                       /* :63-74 */
    n = n + 1;
    orc_expansion(P, S->a, dt_e, &da_1, &da_2);
    S->da = da_1 + da_2;
    ra = S->da / (S->a + S->da);
    if (ra > P->ra_max) dt_e = dt_e * (P->ra_max / ra); else break;
    if (n > 10) break;
  }
  if (P->restrict_da) {                                            /* :76-88 */
    n = 0;
    for (;;) {
      orc_expansion(P, S->a, dt_e, &da_1, &da_2);
      S->da = da_1 + da_2;
      if (S->da > P->da_max) dt_e = dt_e * (P->da_max / S->da); else break;
      n = n + 1;
      if (n > 10) break;
    }
  }
  dt = dt_e;                                                       /* :93-115 */
  if (dt_f_acc < dt) dt = dt_f_acc;
  if ((flags & P3M_FLAG_PPINT) && dt_pp_acc < dt) dt = dt_pp_acc;
  if ((flags & P3M_FLAG_PPINT) && (flags & P3M_FLAG_PP_EXT) && dt_pp_ext_acc < dt) dt = dt_pp_ext_acc;
  if (dt_c_acc < dt) dt = dt_c_acc;
  dt = dt * P->dt_scale;                                           /* :117 */
  orc_expansion(P, S->a, dt, &da_1, &da_2);                        /* :119 */
  S->da = da_1 + da_2;
  S->checkpoint_step = 0; S->projection_step = 0; S->halofind_step = 0;
  {
    float ac = P->a_checkpoint[S->cur_checkpoint - 1], ap = P->a_projection[S->cur_projection - 1], ah = P->a_halofind[S->cur_halofind - 1];
    am = ac; if (ap < am) am = ap; if (ah < am) am = ah;             /* :130 */
    if (ac == am) {                                                /* :135 */
      if (S->a + S->da > ac) {
        S->checkpoint_step = 1;
        dt = dt * (ac - S->a) / S->da;
        orc_expansion(P, S->a, dt, &da_1, &da_2);
        if (S->cur_checkpoint == P->num_checkpoints) S->final_step = 1;
        if (ap == am && S->cur_projection <= P->num_projections) S->projection_step = 1;
        if (ah == am && S->cur_halofind <= P->num_halofinds) S->halofind_step = 1;
      }
    } else if (ap == am && S->cur_projection <= P->num_projections) {   /* :144 */
      if (S->a + S->da > ap) {
        S->projection_step = 1;
        dt = dt * (ap - S->a) / S->da;
        orc_expansion(P, S->a, dt, &da_1, &da_2);
        if (ah == am && S->cur_halofind <= P->num_halofinds) S->halofind_step = 1;
      }
    } else if (ah == am && S->cur_halofind <= P->num_halofinds) {       /* :153 */
      if (S->a + S->da > ah) {
        S->halofind_step = 1;
        dt = dt * (ah - S->a) / S->da;
        orc_expansion(P, S->a, dt, &da_1, &da_2);
      }
    }
  }
  S->dt = dt;
  S->dt_gas = dt / 4;                                              /* :165 */
  S->da = da_1 + da_2;
  S->a_mid = S->a + (S->da / 2);                                   /* :168 */
  S->tau = S->tau + dt; S->t = S->t + dt; S->a = S->a + S->da;     /* :193-195 */
}
