/*
 * oracle/synth_oracle.c — CPU statement of the synthetic G-buffer of SURVEY.md §8(d).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Not from the reference (it has one static Cornell
 * frame and no motion vectors); this is the build's own test scene:
 * 64 seeded axis-aligned rectangles (every 4th one moving) over a background plane, global pan,
 * per-pixel counter-based noise and 5 % fireflies of value 8.  Only +,-,*,compare per pixel, so
 * the GPU generator (csrc/synth.hip) must reproduce it bit for bit.
 */
#include "oracle.h"
#include <math.h>

uint32_t orc_hash32(uint32_t seed, uint32_t frame, uint32_t idx, uint32_t ch)
{
    uint32_t h = seed * 0x9E3779B1u;
    h ^= (frame + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= idx * 0xC2B2AE3Du;
    h ^= (ch + 1u) * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

static float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

typedef struct {
    float cx, cy, hw, hh;
    float n[3];
    float z0, ax, ay;
    float alb[3];
    float shade;
    float ux, uy;
} region_t;

#define NREG 64
#define REGION_FRAME 0xFFFFFFFFu

static void build_regions(uint32_t seed, int W, int H, region_t* R)
{
    const float lx = 0.3f, ly = 0.5f, lz = 0.8f;
    const float linv = 1.0f / sqrtf(lx * lx + ly * ly + lz * lz);
    for (int k = 0; k < NREG; ++k) {
        float u[15];
        for (int j = 0; j < 15; ++j) u[j] = u01(orc_hash32(seed, REGION_FRAME, (uint32_t)k, (uint32_t)j));
        region_t* r = &R[k];
        r->cx = -64.0f + u[0] * (float)(W + 192);
        r->cy = -64.0f + u[1] * (float)(H + 128);
        r->hw = (0.03f + 0.12f * u[2]) * (float)W;
        r->hh = (0.03f + 0.12f * u[3]) * (float)H;
        float dx = 2.0f * u[4] - 1.0f, dy = 2.0f * u[5] - 1.0f, dz = 0.5f + u[6];
        float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        r->n[0] = dx * inv; r->n[1] = dy * inv; r->n[2] = dz * inv;
        r->z0 = 5.0f + 80.0f * u[7];
        r->ax = (u[8] - 0.5f) * 0.02f;
        r->ay = (u[9] - 0.5f) * 0.02f;
        r->alb[0] = 0.2f + 0.7f * u[10]; r->alb[1] = 0.2f + 0.7f * u[11]; r->alb[2] = 0.2f + 0.7f * u[12];
        float ndl = (r->n[0] * lx + r->n[1] * ly + r->n[2] * lz) * linv;
        r->shade = 0.2f + 0.8f * (ndl > 0.0f ? ndl : 0.0f);
        if (k % 4 == 0) {
            r->ux = (floorf(u[13] * 9.0f) - 4.0f) * 0.25f;
            r->uy = (floorf(u[14] * 9.0f) - 4.0f) * 0.25f;
        } else { r->ux = 0.0f; r->uy = 0.0f; }
    }
    region_t* b = &R[NREG];
    b->cx = 0.0f; b->cy = 0.0f; b->hw = 3.0e38f; b->hh = 3.0e38f;
    b->n[0] = 0.0f; b->n[1] = 0.0f; b->n[2] = 1.0f;
    b->z0 = 90.0f; b->ax = 0.0f; b->ay = 0.005f;
    b->alb[0] = b->alb[1] = b->alb[2] = 0.5f;
    b->shade = 0.6f; b->ux = 0.0f; b->uy = 0.0f;
}

void orc_synth_gbuffer(const rmd_synth_desc* d, float* color, float* nd, float* motion, float* albedo)
{
    region_t R[NREG + 1];
    build_regions(d->seed, d->width, d->height, R);
    static const float LIGHT[3] = { 1.0f, 0.95f, 0.9f };
    const int W = d->width;
    const float ff = (float)d->frame;
    for (int r = 0; r < d->buf_rows; ++r) {
        int y = d->buf_row0 + r;
        for (int x = 0; x < W; ++x) {
            size_t o = (size_t)r * W + x;
            float wx = (float)x + ff * d->pan_x, wy = (float)y + ff * d->pan_y;
            int k = NREG;
            float lx = wx, ly = wy;
            for (int j = NREG - 1; j >= 0; --j) {
                float tx = wx - (R[j].cx + ff * R[j].ux), ty = wy - (R[j].cy + ff * R[j].uy);
                if (fabsf(tx) <= R[j].hw && fabsf(ty) <= R[j].hh) { k = j; lx = tx; ly = ty; break; }
            }
            const region_t* g = &R[k];
            float z = g->z0 + g->ax * lx + g->ay * ly;
            if (z < 1.0f) z = 1.0f;
            if (z > 100.0f) z = 100.0f;
            uint32_t idx = (uint32_t)y * (uint32_t)W + (uint32_t)x;
            int firefly = orc_hash32(d->seed, (uint32_t)d->frame, idx, 3u) < 0x0CCCCCCDu;
            for (int ch = 0; ch < 3; ++ch) {
                float u = u01(orc_hash32(d->seed, (uint32_t)d->frame, idx, (uint32_t)ch));
                float c = (g->shade * LIGHT[ch]) * (1.0f + 0.5f * (u - 0.5f));
                color[o * 4 + ch] = firefly ? 8.0f : c;
            }
            color[o * 4 + 3] = 0.0f;
            nd[o * 4 + 0] = g->n[0]; nd[o * 4 + 1] = g->n[1]; nd[o * 4 + 2] = g->n[2]; nd[o * 4 + 3] = z;
            motion[o * 2 + 0] = d->pan_x - g->ux; motion[o * 2 + 1] = d->pan_y - g->uy;
            if (albedo) {
                albedo[o * 4 + 0] = g->alb[0]; albedo[o * 4 + 1] = g->alb[1]; albedo[o * 4 + 2] = g->alb[2];
                albedo[o * 4 + 3] = 1.0f;
            }
        }
    }
}
