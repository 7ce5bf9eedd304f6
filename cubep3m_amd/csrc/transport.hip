// transport.hip -- multi-rank exchanges.  Round-1 state: RCCL entry points are declared and fail
// loudly until the slab transport lands (see DESIGN.md "Multi-GPU").
#include "p3m_internal.h"

extern "C" int p3m_hip_rccl_unique_id(void *unique_id_128) {
  (void)unique_id_128;
  p3m_set_error("RCCL transport not built in this revision");
  return P3M_ECOMM;
}
extern "C" int p3m_hip_comm_init_rccl(p3m_ctx *ctx, const void *unique_id_128) {
  (void)ctx; (void)unique_id_128;
  p3m_set_error("RCCL transport not built in this revision");
  return P3M_ECOMM;
}
