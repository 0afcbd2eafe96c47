/*
 * oracle/svgf_oracle.c — scalar restatement of the SVGF passes.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no temporal / variance / edge-stopping
 * code (README.md:3-10 names the goal; src/filter.cu implements only an unweighted box mean).
 * What the reference does commit to, and this file follows:
 *   - B3-spline taps {3/8, 1/4, 1/16}                     reference src/filter.cu:10
 *   - 5x5 window (radius 2)                               reference src/test.cu:75,87
 *   - tap order dx outer / dy inner                       reference src/filter.cu:34-35
 *   - out-of-range taps skipped, weights renormalised     reference src/filter.cu:38-39,49
 *   - ping-pong between planes per level                  reference src/filter.cu:24-25
 * Everything else is Appendix A (Schied et al. 2017 restated).
 */
#include "oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define PIX4(plane, f, x, y) ((plane) + (((size_t)((y) - (f)->buf_row0)) * (size_t)(f)->width + (size_t)(x)) * 4)
#define PIX2(plane, f, x, y) ((plane) + (((size_t)((y) - (f)->buf_row0)) * (size_t)(f)->width + (size_t)(x)) * 2)
#define PIX1(plane, f, x, y) ((plane) + (((size_t)((y) - (f)->buf_row0)) * (size_t)(f)->width + (size_t)(x)))

static inline float lum3(const float* c) { return 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2]; }
static inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline int is_zero3(const float* n) { return n[0] == 0.0f && n[1] == 0.0f && n[2] == 0.0f; }
static inline float lerpf(float a, float b, float t) { return a + (b - a) * t; }

/* g_z = |z(x+1,y) - z(p)| + |z(x,y+1) - z(p)|, forward differences clamped at the border. */
static inline float depth_gradient(const rmd_svgf_frame_desc* f, const float* nd, int x, int y)
{
    int x1 = x + 1 < f->width ? x + 1 : f->width - 1;
    int y1 = y + 1 < f->height ? y + 1 : f->height - 1;
    float z = PIX4(nd, f, x, y)[3];
    return fabsf(PIX4(nd, f, x1, y)[3] - z) + fabsf(PIX4(nd, f, x, y1)[3] - z);
}

/* Normal edge-stopping weight (Appendix A.2): max(0, n_p.n_t)^sigma_n; both zero => 1,
 * exactly one zero => 0. */
static inline float normal_weight(const float* np, const float* nt, float sigma_n)
{
    int zp = is_zero3(np), zt = is_zero3(nt);
    if (zp || zt) return (zp && zt) ? 1.0f : 0.0f;
    float d = dot3(np, nt);
    if (!(d > 0.0f)) return 0.0f;
    return powf(d, sigma_n);
}

/* ---------------------------------------------------------------------------- T: temporal */
void orc_svgf_temporal(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1)
{
    const int W = f->width, H = f->height;
    static const int TAPX[4] = { 0, 1, 0, 1 }, TAPY[4] = { 0, 0, 1, 1 };
    for (int y = row0; y < row1; ++y) {
        for (int x = 0; x < W; ++x) {
            const float* c  = PIX4(f->color, f, x, y);
            const float* nd = PIX4(f->nd, f, x, y);
            const float* m  = PIX2(f->motion, f, x, y);
            /* A.T.1: q = p + m in fp32, q0 = floor(q), f = q - q0 */
            float qx = (float)x + m[0], qy = (float)y + m[1];
            float fqx = floorf(qx), fqy = floorf(qy);
            int q0x = (int)fqx, q0y = (int)fqy;
            float fx = qx - fqx, fy = qy - fqy;
            float wk[4] = { (1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy };
            /* A.T.2: tap validity */
            float gz = depth_gradient(f, f->nd, x, y);
            float zthr = p->k_z * (gz + 1e-2f);
            int p_zero = is_zero3(nd);
            int mask = 0;
            float wsum = 0.0f, pc[3] = { 0, 0, 0 }, pm1 = 0.0f, pm2 = 0.0f;
            float best_w = -1.0f; int best_h = 0;
            for (int k = 0; k < 4; ++k) {
                if (!f->prev_nd) break;   /* no history: every pixel is a disocclusion */
                int tx = q0x + TAPX[k], ty = q0y + TAPY[k];
                if (tx < 0 || tx >= W || ty < 0 || ty >= H) continue;
                int dyr = ty - y; if (dyr < 0) dyr = -dyr;
                if (dyr > p->max_motion_rows) continue;
                const float* pn = PIX4(f->prev_nd, f, tx, ty);
                if (!(fabsf(pn[3] - nd[3]) <= zthr)) continue;
                int t_zero = is_zero3(pn);
                int ok_n = p_zero ? t_zero : (dot3(pn, nd) >= p->k_n);
                if (!ok_n) continue;
                mask |= 1 << k;
                const float* hc = PIX4(f->hist_color, f, tx, ty);
                const float* hm = PIX2(f->hist_moments, f, tx, ty);      /* float2 (m1, m2) */
                const uint8_t* hl = PIX1(f->hist_len, f, tx, ty);        /* uint8 history length */
                float w = wk[k];
                wsum += w;
                pc[0] += w * hc[0]; pc[1] += w * hc[1]; pc[2] += w * hc[2];
                pm1 += w * hm[0]; pm2 += w * hm[1];
                if (w > best_w) { best_w = w; best_h = (int)hl[0]; }
            }
            /* A.T.3 */
            int h;
            if (mask != 0 && wsum >= 0.01f) {
                pc[0] /= wsum; pc[1] /= wsum; pc[2] /= wsum; pm1 /= wsum; pm2 /= wsum;
                h = best_h + 1; if (h > p->h_max) h = p->h_max; if (h < 1) h = 1;
            } else {
                h = 1;                       /* disocclusion: no history (mask keeps the tap tests) */
                pc[0] = pc[1] = pc[2] = 0.0f; pm1 = pm2 = 0.0f;
            }
            /* A.T.4 */
            float inv_h = 1.0f / (float)h;
            float a_c = p->alpha_color > inv_h ? p->alpha_color : inv_h;
            float a_m = p->alpha_moments > inv_h ? p->alpha_moments : inv_h;
            float l = lum3(c);
            float m1 = lerpf(pm1, l, a_m), m2 = lerpf(pm2, l * l, a_m);
            float var = m2 - m1 * m1; if (!(var > 0.0f)) var = 0.0f;
            float* oc = PIX4(f->t_color, f, x, y);
            float* om = PIX2(f->t_moments, f, x, y);
            oc[0] = lerpf(pc[0], c[0], a_c); oc[1] = lerpf(pc[1], c[1], a_c); oc[2] = lerpf(pc[2], c[2], a_c);
            oc[3] = var;
            om[0] = m1; om[1] = m2;
            *(uint8_t*)PIX1(f->t_len, f, x, y) = (uint8_t)h;          /* 1 <= h <= h_max <= 255 */
            if (f->t_debug) {
                int* od = f->t_debug + (((size_t)(y - f->buf_row0)) * (size_t)W + (size_t)x) * 4;
                od[0] = q0x; od[1] = q0y; od[2] = mask; od[3] = h;
            }
        }
    }
}

/* ---------------------------------------------------------------------------- V: variance */
void orc_svgf_variance(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int row0, int row1)
{
    const int W = f->width, H = f->height, R = p->var_radius;
    for (int y = row0; y < row1; ++y) {
        for (int x = 0; x < W; ++x) {
            const float* c = PIX4(f->t_color, f, x, y);
            float* o = PIX4(f->v_color, f, x, y);
            int h = (int)*PIX1(f->t_len, f, x, y);
            if (h >= p->var_h_threshold) { o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3]; continue; }
            const float* nd = PIX4(f->nd, f, x, y);
            float gz = depth_gradient(f, f->nd, x, y);
            float za = p->sigma_z * (gz > 1e-8f ? gz : 1e-8f);   /* step 1 */
            float sw = 0.0f, sc[3] = { 0, 0, 0 }, sl = 0.0f, sl2 = 0.0f;
            for (int dx = -R; dx <= R; ++dx) {
                for (int dy = -R; dy <= R; ++dy) {
                    int tx = x + dx, ty = y + dy;
                    if (tx < 0 || tx >= W || ty < 0 || ty >= H) continue;
                    const float* tc = PIX4(f->t_color, f, tx, ty);
                    const float* tn = PIX4(f->nd, f, tx, ty);
                    float wn = normal_weight(nd, tn, p->sigma_n);
                    float wz = 0.0f;
                    if (dx != 0 || dy != 0) {
                        float len = sqrtf((float)(dx * dx + dy * dy));
                        wz = fabsf(nd[3] - tn[3]) / (za * len + 1e-8f);
                    }
                    float w = wn * expf(-wz);
                    float tl = lum3(tc);
                    sw += w;
                    sc[0] += w * tc[0]; sc[1] += w * tc[1]; sc[2] += w * tc[2];
                    sl += w * tl; sl2 += w * (tl * tl);
                }
            }
            if (sw < 1e-10f) { o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3]; continue; }
            float el = sl / sw, el2 = sl2 / sw;
            float var = el2 - el * el; if (!(var > 0.0f)) var = 0.0f;
            var *= 4.0f / (float)(h < 1 ? 1 : h);
            o[0] = sc[0] / sw; o[1] = sc[1] / sw; o[2] = sc[2] / sw; o[3] = var;
        }
    }
}

/* ---------------------------------------------------------------------------- A: a-trous */
void orc_svgf_atrous(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int iteration,
                     const float* in, float* out, int row0, int row1)
{
    const int W = f->width, H = f->height;
    const int s = 1 << iteration;
    static const float B3[3] = { 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f };   /* src/filter.cu:10 */
    static const float G3[3] = { 1.0f / 4.0f, 1.0f / 8.0f, 1.0f / 16.0f };   /* A.A.1 prefilter  */
    for (int y = row0; y < row1; ++y) {
        for (int x = 0; x < W; ++x) {
            const float* c = PIX4(in, f, x, y);
            const float* nd = PIX4(f->nd, f, x, y);
            float* o = PIX4(out, f, x, y);
            /* A.A.1: 3x3 Gaussian prefilter of the variance, OOB skipped + renormalised */
            float gsum = 0.0f, vsum = 0.0f;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy) {
                    int tx = x + dx, ty = y + dy;
                    if (tx < 0 || tx >= W || ty < 0 || ty >= H) continue;
                    float g = G3[(dx != 0) + (dy != 0)];
                    gsum += g; vsum += g * PIX4(in, f, tx, ty)[3];
                }
            float var_c = vsum / gsum; if (!(var_c > 0.0f)) var_c = 0.0f;
            float l_den = p->sigma_l * sqrtf(var_c) + 1e-8f;
            float gz = depth_gradient(f, f->nd, x, y);
            float za = p->sigma_z * (gz > 1e-8f ? gz : 1e-8f) * (float)s;
            float lp = lum3(c);
            /* A.A.2 */
            float sw = 0.0f, sc[3] = { 0, 0, 0 }, sv = 0.0f;
            for (int dx = -2; dx <= 2; ++dx) {
                for (int dy = -2; dy <= 2; ++dy) {
                    int tx = x + s * dx, ty = y + s * dy;
                    if (tx < 0 || tx >= W || ty < 0 || ty >= H) continue;
                    const float* tc = PIX4(in, f, tx, ty);
                    const float* tn = PIX4(f->nd, f, tx, ty);
                    float k = B3[dx < 0 ? -dx : dx] * B3[dy < 0 ? -dy : dy];
                    float wn = normal_weight(nd, tn, p->sigma_n);
                    float wz = 0.0f;
                    if (dx != 0 || dy != 0) {
                        float len = sqrtf((float)(dx * dx + dy * dy));
                        wz = fabsf(nd[3] - tn[3]) / (za * len + 1e-8f);
                    }
                    float wl = fabsf(lp - lum3(tc)) / l_den;
                    float w = k * wn * expf(-wz - wl);
                    sw += w;
                    sc[0] += w * tc[0]; sc[1] += w * tc[1]; sc[2] += w * tc[2];
                    sv += (w * w) * tc[3];
                }
            }
            /* A.A.3 */
            if (sw < 1e-10f) { o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3]; continue; }
            o[0] = sc[0] / sw; o[1] = sc[1] / sw; o[2] = sc[2] / sw; o[3] = sv / (sw * sw);
        }
    }
}

/* ------------------------------------------------------------------ threaded drivers */
typedef struct {
    const rmd_svgf_frame_desc* f; const rmd_svgf_params* p;
    int pass, iteration; const float* in; float* out; int row0, row1;
} svgf_job;

static void* svgf_worker(void* a)
{
    svgf_job* j = (svgf_job*)a;
    if (j->pass == 0) orc_svgf_temporal(j->f, j->p, j->row0, j->row1);
    else if (j->pass == 1) orc_svgf_variance(j->f, j->p, j->row0, j->row1);
    else orc_svgf_atrous(j->f, j->p, j->iteration, j->in, j->out, j->row0, j->row1);
    return NULL;
}

void orc_svgf_pass_mt(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int pass, int iteration,
                      const float* in, float* out, int threads)
{
    int r0 = f->buf_row0, r1 = f->buf_row0 + f->buf_rows;
    if (r0 < 0) r0 = 0;
    if (r1 > f->height) r1 = f->height;
    int rows = r1 - r0;
    if (threads < 1) threads = 1;
    if (threads > rows) threads = rows;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    svgf_job* jobs = (svgf_job*)malloc(sizeof(svgf_job) * threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (svgf_job){ f, p, pass, iteration, in, out,
                              r0 + (int)((long)rows * t / threads), r0 + (int)((long)rows * (t + 1) / threads) };
        if (t + 1 < threads) pthread_create(&th[t], NULL, svgf_worker, &jobs[t]);
    }
    svgf_worker(&jobs[threads - 1]);
    for (int t = 0; t + 1 < threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* Plane routing of a whole frame (Appendix A.A.4; same routing as the product's
 * rmd_svgf_frame): V out -> A_0 -> ... ; iteration hist_iteration writes hist_color_out, the
 * last iteration writes out_color, the others ping-pong. */
void orc_svgf_frame(const rmd_svgf_frame_desc* f, const rmd_svgf_params* p, int threads)
{
    orc_svgf_pass_mt(f, p, 0, 0, NULL, NULL, threads);
    orc_svgf_pass_mt(f, p, 1, 0, NULL, NULL, threads);
    const float* in = f->v_color;
    int pp = 0;
    for (int i = 0; i < p->iterations; ++i) {
        float* out;
        if (i == p->iterations - 1) out = f->out_color;
        else if (i == p->hist_iteration) out = f->hist_color_out;
        else { out = f->ping[pp]; pp ^= 1; }
        orc_svgf_pass_mt(f, p, 2, i, in, out, threads);
        if (i == p->iterations - 1 && i == p->hist_iteration && f->hist_color_out && f->hist_color_out != out)
            memcpy(f->hist_color_out, out, (size_t)f->buf_rows * f->width * 4 * sizeof(float));
        in = out;
    }
}

/* ------------------------------------------------------------------ 8-bit conversions */
void orc_convert_u8_to_f32(const uint8_t* in, float* out, size_t pixels, int renormalize_xyz, float w_value)
{
    for (size_t i = 0; i < pixels; ++i) {
        float v[3] = { (float)in[i * 4 + 0] / 255.0f, (float)in[i * 4 + 1] / 255.0f, (float)in[i * 4 + 2] / 255.0f };
        if (renormalize_xyz) {
            float l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            if (l2 > 0.0f) { float inv = 1.0f / sqrtf(l2); v[0] *= inv; v[1] *= inv; v[2] *= inv; }
        }
        out[i * 4 + 0] = v[0]; out[i * 4 + 1] = v[1]; out[i * 4 + 2] = v[2];
        out[i * 4 + 3] = (w_value < 0.0f) ? (float)in[i * 4 + 3] / 255.0f : w_value;
    }
}

/* Demodulation by albedo: illumination = radiance / max(albedo, eps), w unchanged. */
void orc_demodulate(const float* radiance, const float* albedo, float* out, size_t pixels, float eps)
{
    for (size_t i = 0; i < pixels; ++i) {
        for (int ch = 0; ch < 3; ++ch) {
            float a = albedo[i * 4 + ch];
            if (!(a > eps)) a = eps;
            out[i * 4 + ch] = radiance[i * 4 + ch] / a;
        }
        out[i * 4 + 3] = radiance[i * 4 + 3];
    }
}

void orc_convert_f32_to_u8(const float* in, const float* albedo, uint8_t* out, size_t pixels)
{
    for (size_t i = 0; i < pixels; ++i) {
        for (int ch = 0; ch < 3; ++ch) {
            float v = in[i * 4 + ch];
            if (albedo) v = v * albedo[i * 4 + ch];
            v = v * 255.0f + 0.5f;
            if (!(v > 0.0f)) v = 0.0f;
            if (v > 255.0f) v = 255.0f;
            out[i * 4 + ch] = (uint8_t)v;
        }
        out[i * 4 + 3] = 255;
    }
}

/* The product's 8-bit front end (csrc/pixel_convert.h unit_from_u8) computes (float)b / 255.0f WITHOUT a division: q = v * RN(1/255),
 * e = fma(-q, 255, v) (exact residual), result = fma(e, RN(1/255), q).  This restates that formula with C99 fmaf and counts the
 * bytes for which it differs from the IEEE quotient the oracle (and the reference-style c/255) uses: must be 0
 * (tests/test_gbuffer_frame.py).  Test infrastructure, like everything in this directory. */
int orc_unit_from_u8_mismatches(void)
{
    const float r = 1.0f / 255.0f;
    int bad = 0;
    for (int b = 0; b < 256; ++b) {
        const float v = (float)b;
        const float q = v * r;
        const float e = fmaf(-q, 255.0f, v);
        if (fmaf(e, r, q) != v / 255.0f) ++bad;
    }
    return bad;
}

