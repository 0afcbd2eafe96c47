/*
 * oracle/abi_probe.c — the layout of every struct of include/rmd_api.h as THIS C compiler sees it.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * tests/oracle_lib.py builds its ctypes structures from this text instead of importing them from the
 * product's Python binding (raymarchdenoisercuda_amd/_lib.py): a wrong field order there would
 * otherwise be wrong identically on both sides of every parity test.  tests/test_abi.py compares the
 * binding with it field by field.  The two structs the reference defines are cited where they
 * come from: rmd_gbuffer = reference include/gbuffer.h:6-14, rmd_filter_params = include/filter.cuh:11-23.
 *
 * Line format:  "<struct> <sizeof> <alignof>"  then  " <field> <offset> <size> <kind>"  per field,
 * kind = i (int) | u (uint32) | f (float) | b (unsigned char) | p (pointer) | s:<struct> (nested).
 */
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "oracle.h"

#define BEGIN(S)        n += snprintf(buf + n, sizeof(buf) - n, "%s %zu %zu\n", #S, sizeof(S), __alignof__(S))
#define FIELD(S, F, K)  n += snprintf(buf + n, sizeof(buf) - n, " %s %zu %zu %s\n", #F, offsetof(S, F), sizeof(((S*)0)->F), K)

const char* orc_abi_layout(void)
{
    static char buf[8192];
    size_t n = 0;
    BEGIN(rmd_int2);
    FIELD(rmd_int2, x, "i"); FIELD(rmd_int2, y, "i");
    BEGIN(rmd_gbuffer);                                 /* reference include/gbuffer.h:6-14 */
    FIELD(rmd_gbuffer, shape, "s:rmd_int2"); FIELD(rmd_gbuffer, render, "p"); FIELD(rmd_gbuffer, denoised, "p");
    FIELD(rmd_gbuffer, normal, "p"); FIELD(rmd_gbuffer, albedo, "p"); FIELD(rmd_gbuffer, buffer, "p");
    BEGIN(rmd_filter_params);                           /* reference include/filter.cuh:11-23 */
    FIELD(rmd_filter_params, type, "i"); FIELD(rmd_filter_params, depth, "i"); FIELD(rmd_filter_params, level, "i");
    FIELD(rmd_filter_params, radius, "i"); FIELD(rmd_filter_params, sigmaSpace, "f"); FIELD(rmd_filter_params, sigmaColor, "f");
    FIELD(rmd_filter_params, sigmaAlbedo, "f"); FIELD(rmd_filter_params, sigmaNormal, "f");
    FIELD(rmd_filter_params, cacheInput, "b"); FIELD(rmd_filter_params, cacheBuffer, "b");
    BEGIN(rmd_svgf_params);
    FIELD(rmd_svgf_params, alpha_color, "f"); FIELD(rmd_svgf_params, alpha_moments, "f"); FIELD(rmd_svgf_params, h_max, "i");
    FIELD(rmd_svgf_params, k_z, "f"); FIELD(rmd_svgf_params, k_n, "f"); FIELD(rmd_svgf_params, max_motion_rows, "i");
    FIELD(rmd_svgf_params, var_h_threshold, "i"); FIELD(rmd_svgf_params, var_radius, "i");
    FIELD(rmd_svgf_params, sigma_n, "f"); FIELD(rmd_svgf_params, sigma_z, "f"); FIELD(rmd_svgf_params, sigma_l, "f");
    FIELD(rmd_svgf_params, iterations, "i"); FIELD(rmd_svgf_params, hist_iteration, "i"); FIELD(rmd_svgf_params, atrous_variant, "i");
    FIELD(rmd_svgf_params, tv_workgroups, "i"); FIELD(rmd_svgf_params, atrous_cus, "i"); FIELD(rmd_svgf_params, exchange_iteration, "i");
    BEGIN(rmd_svgf_frame_desc);
    FIELD(rmd_svgf_frame_desc, width, "i"); FIELD(rmd_svgf_frame_desc, height, "i");
    FIELD(rmd_svgf_frame_desc, buf_row0, "i"); FIELD(rmd_svgf_frame_desc, buf_rows, "i");
    FIELD(rmd_svgf_frame_desc, color, "p"); FIELD(rmd_svgf_frame_desc, nd, "p"); FIELD(rmd_svgf_frame_desc, motion, "p");
    FIELD(rmd_svgf_frame_desc, hist_color, "p"); FIELD(rmd_svgf_frame_desc, hist_moments, "p"); FIELD(rmd_svgf_frame_desc, hist_len, "p");
    FIELD(rmd_svgf_frame_desc, prev_nd, "p");
    FIELD(rmd_svgf_frame_desc, t_color, "p"); FIELD(rmd_svgf_frame_desc, t_moments, "p"); FIELD(rmd_svgf_frame_desc, t_len, "p");
    FIELD(rmd_svgf_frame_desc, t_debug, "p");
    FIELD(rmd_svgf_frame_desc, v_color, "p"); FIELD(rmd_svgf_frame_desc, hist_color_out, "p"); FIELD(rmd_svgf_frame_desc, ping, "p");
    FIELD(rmd_svgf_frame_desc, out_color, "p"); FIELD(rmd_svgf_frame_desc, stats, "p"); FIELD(rmd_svgf_frame_desc, v_tile_flags, "p");
    BEGIN(rmd_strip_plan);
    FIELD(rmd_strip_plan, height, "i"); FIELD(rmd_strip_plan, world, "i"); FIELD(rmd_strip_plan, rank, "i");
    FIELD(rmd_strip_plan, row0, "i"); FIELD(rmd_strip_plan, row1, "i"); FIELD(rmd_strip_plan, buf_row0, "i");
    FIELD(rmd_strip_plan, buf_rows, "i"); FIELD(rmd_strip_plan, reach_in, "i"); FIELD(rmd_strip_plan, reach_hist, "i");
    FIELD(rmd_strip_plan, have_color, "i"); FIELD(rmd_strip_plan, have_moments, "i");
    FIELD(rmd_strip_plan, mid_iteration, "i"); FIELD(rmd_strip_plan, mid_rows, "i");
    BEGIN(rmd_halo_step);
    FIELD(rmd_halo_step, kind, "i"); FIELD(rmd_halo_step, plane, "i"); FIELD(rmd_halo_step, row_lo, "i");
    FIELD(rmd_halo_step, row_hi, "i"); FIELD(rmd_halo_step, peer, "i");
    BEGIN(rmd_synth_desc);
    FIELD(rmd_synth_desc, width, "i"); FIELD(rmd_synth_desc, height, "i"); FIELD(rmd_synth_desc, buf_row0, "i");
    FIELD(rmd_synth_desc, buf_rows, "i"); FIELD(rmd_synth_desc, seed, "u"); FIELD(rmd_synth_desc, frame, "i");
    FIELD(rmd_synth_desc, pan_x, "f"); FIELD(rmd_synth_desc, pan_y, "f");
    (void)n;
    return buf;
}
