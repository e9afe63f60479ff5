/*
 * mm_counts.h -- vehicle counts of an episode (MergeEnv._num_vehicles, highway_env/envs/merge_env_v1.py:180-211, and its
 * MergeEnvLCMARL override :476-495) as pure arithmetic on the two uniform draws, plus the capacity check that the
 * reference leaves to np.random.choice(..., replace=False) raising ValueError (merge_env_v1.py:284-320: six spawn points
 * per road, shared by the road's CAVs and HDVs).  Shared by the HIP library and the CPU oracle so that the device-side
 * count draw, its oracle twin and the configuration check cannot drift apart.
 *
 * MMConfig.mixed_traffic: 0 = CAVs only (the drawn HDV count is added to the CAVs), 1 = mixed, 2 = traffic_type "av"
 * (one CAV, everything else HDVs).  MMConfig.num_cav > 0 = reset(num_CAV=k) override of the CAV draw.
 */
#ifndef MM_COUNTS_H
#define MM_COUNTS_H

#include <stdio.h>

#include "mm_abi.h"

#ifndef MM_COUNTS_FN
#define MM_COUNTS_FN static inline /* the arithmetic (a HIP includer makes it __host__ __device__) */
#endif
#ifndef MM_COUNTS_HOST_FN
#define MM_COUNTS_HOST_FN static inline /* the configuration check (host only: it formats a message) */
#endif

/* i_cav, i_hdv in 0..2: the index np.random.choice(np.arange(lo, lo + 3), 1) drew */
MM_COUNTS_FN void mm_counts_from_draw(int traffic_density, int mixed_traffic, int num_cav, int i_cav, int i_hdv, int *n_cav,
                                      int *n_hdv) {
  const int lo_c = traffic_density == 1 ? 1 : (traffic_density == 2 ? 2 : 4);
  const int lo_h = traffic_density == 1 ? 1 : (traffic_density == 2 ? 2 : 3);
  int nc = num_cav > 0 ? num_cav : lo_c + i_cav;
  int nh = lo_h + i_hdv;
  if (mixed_traffic == 0) { nc = nc + nh; nh = 0; }               /* :206-209 */
  else if (mixed_traffic == 2) { nh = nc + nh - 1; nc = 1; }       /* traffic_type "av" :485-489 */
  *n_cav = nc; *n_hdv = nh;
}

/* spawn points a composition needs per road in the worst case (a single vehicle of a kind goes to either road) */
MM_COUNTS_FN void mm_counts_road_need(int n_cav, int n_hdv, int *main_road, int *ramp) {
  const int cs = n_cav != 1 ? n_cav / 2 : 1, cm = n_cav != 1 ? n_cav - n_cav / 2 : 1;
  const int hs = n_hdv != 1 ? n_hdv / 2 : 1, hm = n_hdv != 1 ? n_hdv - n_hdv / 2 : 1;
  *main_road = cs + hs; *ramp = cm + hm;
}

/* 0 when every composition the configuration can produce fits N slots and six spawn points per road; else writes why.
 * fixed_too = 0 (mm_create / mm_set_config): only the per-episode draw is checked -- with fixed counts the vehicles may come
 * from the host (mm_init_from_kinematics) in any number up to N, and an auto-reset re-spawns what the env holds;
 * fixed_too = 1 (mm_reset): the device spawns N - n_hdv CAVs + n_hdv HDVs itself. */
MM_COUNTS_HOST_FN int mm_counts_check(const MMConfig *c, int N, int fixed_too, char *err, size_t err_len) {
  if (c->traffic_density < 0 || c->traffic_density > 3) { snprintf(err, err_len, "traffic_density %d is not 0..3", c->traffic_density); return 1; }
  if (c->num_cav < 0) { snprintf(err, err_len, "num_cav %d is negative", c->num_cav); return 1; }
  if (c->mixed_traffic < 0 || c->mixed_traffic > 2) { snprintf(err, err_len, "mixed_traffic %d is not 0 (cav) / 1 (mixed) / 2 (av)", c->mixed_traffic); return 1; }
  if (c->traffic_density == 0 && !fixed_too) return 0;
  for (int ic = 0; ic < 3; ic++)
    for (int ih = 0; ih < 3; ih++) {
      int nc = N - c->n_hdv, nh = c->n_hdv, ms, mm;
      if (c->traffic_density > 0) mm_counts_from_draw(c->traffic_density, c->mixed_traffic, c->num_cav, ic, ih, &nc, &nh);
      if (nc + nh > N) {
        snprintf(err, err_len, "traffic_density %d%s can draw %d CAVs + %d HDVs, the batch has %d slots per env", c->traffic_density,
                 c->num_cav > 0 ? " with the num_CAV override" : "", nc, nh, N);
        return 1;
      }
      mm_counts_road_need(nc, nh, &ms, &mm);
      if (ms > 6 || mm > 6) {
        snprintf(err, err_len, "%d CAVs + %d HDVs need %d / %d spawn points on the main road / ramp, each has 6 "
                 "(the reference's np.random.choice(replace=False) raises here)", nc, nh, ms, mm);
        return 1;
      }
    }
  return 0;
}

#endif /* MM_COUNTS_H */
