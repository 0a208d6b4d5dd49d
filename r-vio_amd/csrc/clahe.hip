// clahe.hip — T0: cv::createCLAHE(3.0, Size(5,5))->apply(im, im)   (Tracker.cc:198-202, enable_equalizer = 1).
// OpenCV's imgproc/src/clahe.cpp (CLAHE_CalcLut_Body + CLAHE_Interpolation_Body, 8-bit path) restated for the device;
// bit-exact against oracle/frontend.cpp:clahe_apply (integer histogram work; float blend with contraction off and
// round-half-even, the two roundings OpenCV's cvRound performs).  Included inside the fp-contract(off) region.
//   clahe_lut_kernel    one workgroup per tile: per-wave LDS histograms (reflect-101 extension to the right/bottom),
//                       clip, redistribute, cumulative sum (DPP scan), LUT
//   clahe_interp_kernel one thread per pixel: bilinear blend of the four neighbouring tile LUTs (LUTs staged in LDS)
#pragma once

#define CLAHE_LUT_T 1024
__global__ __launch_bounds__(CLAHE_LUT_T) void clahe_lut_kernel(const uint8_t* __restrict__ src, int w, int h, int stride, int tiles_x, int tw, int th,
                                                                int clip_limit, float lut_scale, uint8_t* __restrict__ lut, size_t src_bs, size_t bs, int dbg_tag) {
    DBG_I(blockIdx.x == 0 && blockIdx.z == 0, dbg_tag, 0);
    src = zoff(src, src_bs); lut = zoff(lut, bs);
    __shared__ int hist[16][256];
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int t = blockIdx.x, ty = t / tiles_x, tx = t % tiles_x;
#pragma unroll
    for (int k = 0; k < 4; ++k) hist[4 * k + (tid >> 8)][tid & 255] = 0;
    __syncthreads();
    // wave <-> every 16th row of the tile, two rows x 256 columns of byte loads in flight before the LDS atomics
    for (int r = wv; r < th; r += 32) {
        const uint8_t* row0 = src + (size_t)reflect1(ty * th + r, h) * stride;
        const bool has1 = r + 16 < th;
        const uint8_t* row1 = src + (size_t)reflect1(ty * th + (has1 ? r + 16 : r), h) * stride;
        for (int c0 = 0; c0 < tw; c0 += 256) {
            int v0[4], v1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + lane + 64 * j;
                const int xs = reflect1(tx * tw + (c < tw ? c : 0), w);
                v0[j] = c < tw ? (int)row0[xs] : -1;
                v1[j] = (c < tw && has1) ? (int)row1[xs] : -1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (v0[j] >= 0) atomicAdd(&hist[wv][v0[j]], 1);
                if (v1[j] >= 0) atomicAdd(&hist[wv][v1[j]], 1);
            }
        }
    }
    __syncthreads();
    int hv = 0;
    if (tid < 256) {
#pragma unroll
        for (int k = 0; k < 16; ++k) hv += hist[k][tid];
    }
    if (clip_limit > 0) {
        const int excess = hv > clip_limit ? hv - clip_limit : 0;
        hv -= excess;
        int clipped;
        block_exscan(excess, &clipped, s_w);
        const int batch = clipped / 256;
        const int residual = clipped - batch * 256;
        hv += batch;
        if (residual != 0) {
            const int step = 256 / residual > 1 ? 256 / residual : 1;
            if (tid % step == 0 && tid / step < residual) hv += 1;
        }
    }
    int total;
    const int sum = block_exscan(hv, &total, s_w) + hv;      // inclusive
    int v = (int)rintf((float)sum * lut_scale);
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    if (tid < 256) lut[(size_t)t * 256 + tid] = (uint8_t)v;
}

// Round 4: the histogram without LDS-atomic conflicts.  clahe_lut_kernel lets the 64 lanes of a wave add into ONE 256-bin histogram: an image
// region has few grey levels, so most lanes of an instruction hit the same bins and the LDS serialises them (731 us per batched frame of 128
// streams — 63 GB/s of image bytes —, 12.7 us for one).  Here every LANE owns a column of the histogram: hist[bin][lane] as 16-bit counters
// packed in pairs (a counter sees one lane column of the tile, ceil(tw / 64) th pixels <= 65535; 256 x 32 words = 32 KB, five workgroups per CU), so an instruction's 64 additions go to
// 64 different counters (two lanes share a word: a two-way conflict at worst), and the four waves of the workgroup meet only by coincidence.
// The counts are integers: the same histogram, hence the same LUT bit for bit.  256 threads: wave <-> every fourth row of the tile.
// T threads: 256 for batch handles (five workgroups per CU), 1024 for one stream (the tile's latency: 16 waves share the rows).
template <int T, int RIF = 4>      // RIF: rows of byte loads in flight per thread before the LDS additions
__global__ __launch_bounds__(T) void clahe_lut_kernel2(const uint8_t* __restrict__ src, int w, int h, int stride, int tiles_x, int tw, int th,
                                                       int clip_limit, float lut_scale, uint8_t* __restrict__ lut, size_t src_bs, size_t bs, int dbg_tag) {
    DBG_I(blockIdx.x == 0 && blockIdx.z == 0, dbg_tag, 0);
    src = zoff(src, src_bs); lut = zoff(lut, bs);
    constexpr int NWV = T / 64;
    __shared__ unsigned hist2[256 * 32];
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int t = blockIdx.x, ty = t / tiles_x, tx = t % tiles_x;
#pragma unroll
    for (int k = 0; k < 256 * 32 / T; ++k) hist2[tid + T * k] = 0u;
    __syncthreads();
    const unsigned one = 1u << (16 * (lane & 1));
    unsigned* const mine = hist2 + (lane >> 1);
    // RIF rows x 64 columns of byte loads in flight before the LDS additions; wave <-> every NWV-th row
    for (int c0 = 0; c0 < tw; c0 += 64) {
        const int c = c0 + lane;
        const bool cok = c < tw;
        const int xs = reflect1(tx * tw + (cok ? c : 0), w);
        for (int r0 = wv; r0 < th; r0 += RIF * NWV) {
            int v[RIF];
#pragma unroll
            for (int j = 0; j < RIF; ++j) {
                const int r = r0 + NWV * j;
                v[j] = (cok && r < th) ? (int)src[(size_t)reflect1(ty * th + r, h) * stride + xs] : -1;
            }
#pragma unroll
            for (int j = 0; j < RIF; ++j) if (v[j] >= 0) atomicAdd(&mine[v[j] * 32], one);
        }
    }
    __syncthreads();
    int hv = 0;
    if (tid < 256) {   // bin tid: the sum of its 64 counters (the 32 words read in a rotated order: consecutive bins sit 32 words apart)
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) { const unsigned q = hist2[tid * 32 + ((k + tid) & 31)]; lo += q & 0xffffu; hi += q >> 16; }
        hv = (int)(lo + hi);
    }
    if (clip_limit > 0) {
        const int excess = hv > clip_limit ? hv - clip_limit : 0;
        hv -= excess;
        int clipped;
        block_exscan(excess, &clipped, s_w);
        const int batch = clipped / 256;
        const int residual = clipped - batch * 256;
        hv += batch;
        if (residual != 0) {
            const int step = 256 / residual > 1 ? 256 / residual : 1;
            if (tid % step == 0 && tid / step < residual) hv += 1;
        }
    }
    int total;
    const int sum = block_exscan(hv, &total, s_w) + hv;      // inclusive
    int v = (int)rintf((float)sum * lut_scale);
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    if (tid < 256) lut[(size_t)t * 256 + tid] = (uint8_t)v;
}

#define CLAHE_MAX_TILES 64
__global__ __launch_bounds__(256) void clahe_interp_kernel(const uint8_t* __restrict__ src, int w, int h, int stride, int tiles_x, int tiles_y,
                                                           float inv_tw, float inv_th, const uint8_t* __restrict__ lut, uint8_t* __restrict__ dst,
                                                           size_t src_bs, size_t bs) {
    src = zoff(src, src_bs); lut = zoff(lut, bs); dst = zoff(dst, bs);
    __shared__ uint32_t s_lut[CLAHE_MAX_TILES * 64];
    const int tid = threadIdx.x;
    const int nwords = tiles_x * tiles_y * 64;
    const uint32_t* lut32 = (const uint32_t*)lut;
    for (int e = tid; e < nwords; e += 256) s_lut[e] = lut32[e];
    __syncthreads();
    const uint8_t* L = (const uint8_t*)s_lut;
    const int x = blockIdx.x * 64 + (tid & 63), y = blockIdx.y * 4 + (tid >> 6);
    if (x >= w || y >= h) return;
    const float tyf = (float)y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
    ty1 = ty1 > 0 ? ty1 : 0; ty2 = ty2 < tiles_y - 1 ? ty2 : tiles_y - 1;
    const float txf = (float)x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
    tx1 = tx1 > 0 ? tx1 : 0; tx2 = tx2 < tiles_x - 1 ? tx2 : tiles_x - 1;
    const int v = src[(size_t)y * stride + x];
    const uint8_t* p1 = L + ty1 * tiles_x * 256;
    const uint8_t* p2 = L + ty2 * tiles_x * 256;
    const int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
    const float res = ((float)p1[i1] * xa1 + (float)p1[i2] * xa) * ya1 + ((float)p2[i1] * xa1 + (float)p2[i2] * xa) * ya;
    int r = (int)rintf(res);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    dst[(size_t)y * w + x] = (uint8_t)r;
}

// Throughput form (batched launches): one workgroup = 256 x 16 pixels, one thread = 4 adjacent pixels of 4 rows, so the tile LUTs
// (6.4 KB) are staged once per 4096 pixels instead of once per 256.  Same per-pixel expression.  Requires w % 4 == 0,
// stride % 4 == 0 and word-aligned rows (host-checked).
__global__ __launch_bounds__(256) void clahe_interp_kernel4(const uint8_t* __restrict__ src, int w, int h, int stride, int tiles_x, int tiles_y,
                                                            float inv_tw, float inv_th, const uint8_t* __restrict__ lut, uint8_t* __restrict__ dst,
                                                            size_t src_bs, size_t bs) {
    src = zoff(src, src_bs); lut = zoff(lut, bs); dst = zoff(dst, bs);
    __shared__ uint32_t s_lut[CLAHE_MAX_TILES * 64];
    const int tid = threadIdx.x;
    const int nwords = tiles_x * tiles_y * 64;
    const uint32_t* lut32 = (const uint32_t*)lut;
    for (int e = tid; e < nwords; e += 256) s_lut[e] = lut32[e];
    __syncthreads();
    const uint8_t* L = (const uint8_t*)s_lut;
    const int x = (blockIdx.x * 64 + (tid & 63)) * 4;
    if (x >= w) return;
    int tx1[4], tx2[4];
    float xa[4], xa1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float txf = (float)(x + k) * inv_tw - 0.5f;
        int t1 = (int)floorf(txf), t2 = t1 + 1;
        xa[k] = txf - (float)t1; xa1[k] = 1.0f - xa[k];
        tx1[k] = (t1 > 0 ? t1 : 0) * 256; tx2[k] = (t2 < tiles_x - 1 ? t2 : tiles_x - 1) * 256;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int y = blockIdx.y * 16 + rr * 4 + (tid >> 6);
        if (y >= h) continue;
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        ty1 = ty1 > 0 ? ty1 : 0; ty2 = ty2 < tiles_y - 1 ? ty2 : tiles_y - 1;
        const uint8_t* p1 = L + ty1 * tiles_x * 256;
        const uint8_t* p2 = L + ty2 * tiles_x * 256;
        const uint32_t v4 = *(const uint32_t*)(src + (size_t)y * stride + x);
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = (v4 >> (8 * k)) & 255;
            const int i1 = tx1[k] + v, i2 = tx2[k] + v;
            const float res = ((float)p1[i1] * xa1[k] + (float)p1[i2] * xa[k]) * ya1 + ((float)p2[i1] * xa1[k] + (float)p2[i2] * xa[k]) * ya;
            int r = (int)rintf(res);
            r = r < 0 ? 0 : (r > 255 ? 255 : r);
            out |= (uint32_t)r << (8 * k);
        }
        *(uint32_t*)(dst + (size_t)y * w + x) = out;
    }
}
