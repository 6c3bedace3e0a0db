/*
 * orb_describe.hip -- intensity-centroid orientation, steered BRIEF-256 and output assembly: ONE
 * wavefront per keypoint.
 *
 * Replaces computeOrientation/IC_Angle (reference src/ORBextractor.cpp:27-54, :422-429),
 * computeOrbDescriptor (:58-97), the post-distribute fix-ups (:795-805) and the epilogue of
 * operator() (:1019-1064: level-order concatenation, pt *= mvScaleFactor[level]).
 *
 *   - orientation: lanes 0..30 own the 31 columns u=-15..15 of the circular patch and walk its rows
 *     (31 contiguous bytes per row: one coalesced segment), integer moments are reduced across the
 *     wavefront with shuffles, cv::fastAtan2's f32 polynomial gives the angle;
 *   - descriptor: lane L evaluates bits L, L+64, L+128, L+192 (two rotated u8 gathers each from the
 *     BLURRED level); a wavefront ballot of each group IS 8 consecutive descriptor bytes;
 *   - the rotation uses a deterministic double-precision sin/cos (same definition as the oracle's
 *     ORC_TRIG_DET) and f32 mul/add without FMA contraction, then round-half-even (cvRound).
 */
#include "orb_common.h"
#include <mutex>

__constant__ int8_t c_pattern[1024]; /* 256 x (x0,y0,x1,y1), reference src/ORBextractor.cpp:100-358 */
static const int8_t h_pattern[1024] = {
#include "orb_pattern_data.inc"
};
/* circular patch half-widths, derived by the reference ctor (:404-419) */
__constant__ int8_t c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* deterministic sin/cos: Cody-Waite reduction + fdlibm kernels in plain IEEE double (no FMA) */
__device__ __forceinline__ void sincos_det(float angle, float *c_out, float *s_out) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)angle;
    double kd = rint(x * two_over_pi);
    int k = (int)kd;
    double r = (x - kd * pio2_hi) - kd * pio2_lo;
    double z = r * r;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sn = r + (r * z) * ps;
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    double sv, cv;
    switch (k & 3) {
    case 0: sv = sn; cv = cs; break;
    case 1: sv = cs; cv = -sn; break;
    case 2: sv = -sn; cv = -cs; break;
    default: sv = -cs; cv = sn; break;
    }
    *c_out = (float)cv;
    *s_out = (float)sv;
}

/* One workgroup = 4 wavefronts x SPW keypoint slots, in three phases that share LDS:
 *   1. (wavefront per slot) intensity-centroid moments.  The 31 x 31 patch is read as 31 rows x 8 (unaligned)
 *      dwords, four per lane, and the circular mask and the column weights ride in two byte-weight dwords per
 *      (row, dword): row sum S0 = dot4(I, mask), S1 = dot4(I, (u + 15) * mask); m01 = sum v * S0,
 *      m10 = sum S1 - 15 * sum S0.  Two v_dot4_u32_u8 per dword where a column-per-lane walk spent ~10
 *      instructions per row on HALF the lanes.
 *   2. (one LANE per slot) cv::fastAtan2 and the deterministic double-precision sin / cos.  They are ~90 vector
 *      instructions -- two dozen in f64 -- which a wavefront-per-keypoint kernel issues once PER KEYPOINT for 64
 *      identical lanes; here the 4 x SPW slots of the workgroup share one issue of them.
 *   3. (wavefront per slot) steered BRIEF: lane L evaluates bits L, L + 64, L + 128, L + 192; a ballot of each group
 *      is 8 descriptor bytes.
 * SPW = 4 for batches (see launch_describe), SPW = 1 for calls of a few frames, where a frame's thousand keypoints
 * must spread over the whole device rather than queue up in a few workgroups.
 * Integer moments are exact in any order and the float / double steps are the same functions on the same inputs:
 * bit-identical to the previous kernel and to the oracle. */
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
template <int SPW>
__global__ __launch_bounds__(256) void k_describe(OrbGeom g, const uint8_t *pyr, const uint8_t *blur,
                                                   const uint32_t *lvl_kp, const int32_t *lvl_cnt, OrbKeyPoint *kps,
                                                   uint8_t *desc, int cap, int32_t *counts) {
    constexpr int SLOTS = 4 * SPW;
    __shared__ uint32_t s_w[248 * 2]; /* per (row, dword) of the patch: mask bytes, (u + 15) * mask bytes */
    __shared__ int s_lvl[SLOTS], s_oi[SLOTS], s_m01[SLOTS], s_m10[SLOTS];
    __shared__ uint32_t s_key[SLOTS];
    __shared__ float s_ang[SLOTS], s_a[SLOTS], s_b[SLOTS];
    /* all keypoints of one frame run on ONE XCD (consecutive remapped items share an L2): the frame's pyramid and
       blurred pyramid (2 MB) then stay L2-resident while its ~1000 patches are gathered from them */
    const int item = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int frame = item / (int)gridDim.x, bx = item - frame * (int)gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int32_t *lc = lvl_cnt + (int64_t)frame * g.nlevels;
    if (tid < 248) {
        const int row = tid >> 3, dw = tid & 7, v = row - 15, um = c_umax[v < 0 ? -v : v];
        uint32_t mw = 0, uw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int u = dw * 4 + k - 15;
            if (u >= -um && u <= um) {
                mw |= 1u << (8 * k);
                uw |= (uint32_t)(u + 15) << (8 * k);
            }
        }
        s_w[2 * tid] = mw;
        s_w[2 * tid + 1] = uw;
    }
    if (tid < SLOTS) { /* slot -> (level, position in the concatenated output (:1035-1063)) */
        const int slot = bx * SLOTS + tid;
        int level = 0, before = 0, cnt = lc[0], prefix = 0;
        for (int l = 1; l < g.nlevels; l++) {
            prefix += lc[l - 1];
            if (slot >= g.lv[l].out_base) {
                level = l;
                before = prefix;
                cnt = lc[l];
            }
        }
        int base = 0;
        for (int l = 1; l < g.nlevels; l++) base = slot >= g.lv[l].out_base ? g.lv[l].out_base : base;
        const int idx = slot - base, oi = before + idx;
        const bool valid = slot < g.total_out_cap && idx < cnt && oi < cap;
        s_lvl[tid] = valid ? level : -1;
        s_oi[tid] = oi;
        s_key[tid] = valid ? lvl_kp[(int64_t)frame * g.total_out_cap + slot] : 0u;
    }
    if (bx == 0 && tid == 0) {
        int tot = 0;
        for (int l = 0; l < g.nlevels; l++) tot += lc[l];
        counts[frame] = tot < cap ? tot : cap;
    }
    __syncthreads();
    /* ---- 1. IC_Angle moments on the un-blurred level ---- */
    for (int k = 0; k < SPW; k++) {
        const int sl = wave * SPW + k;
        const int level = __builtin_amdgcn_readfirstlane(s_lvl[sl]);
        if (level < 0) continue;
        /* the keypoint is the same for the whole wavefront: it (and every pointer derived from it) stays in SGPRs */
        const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_key[sl]);
        const LevelGeom &L = g.lv[level];
        const int px = KEY_X(key) + ORB_MIN_BORDER, py = KEY_Y(key) + ORB_MIN_BORDER; /* :799-800 */
        const int pitch = L.pitch;
        const uint8_t *patch = level_ptr(pyr, g, frame, level) + (int64_t)(py - 15) * pitch + (px - 15); /* top-left of the 31 x 31 patch */
        uint32_t I[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { /* the 31 x 32-byte window lies inside the image: EDGE_THRESHOLD 19 > 16 */
            const int i = min(lane + 64 * j, 247);
            I[j] = *(const u32_unaligned *)(patch + (uint32_t)(__mul24(i >> 3, pitch) + (i & 7) * 4));
        }
        int m01 = 0;
        uint32_t s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = lane + 64 * j;
            if (i < 248) {
                const uint32_t r0 = __builtin_amdgcn_udot4(I[j], s_w[2 * i], 0u, false);
                s1 = __builtin_amdgcn_udot4(I[j], s_w[2 * i + 1], s1, false);
                s0 += r0;
                m01 += ((i >> 3) - 15) * (int)r0;
            }
        }
        int m10 = (int)s1 - 15 * (int)s0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m10 += __shfl_xor(m10, o);
            m01 += __shfl_xor(m01, o);
        }
        if (lane == 0) {
            s_m01[sl] = m01;
            s_m10[sl] = m10;
        }
    }
    __syncthreads();
    /* ---- 2. angle and rotation, one lane per slot ---- */
    if (tid < SLOTS && s_lvl[tid] >= 0) {
        const float angle = fast_atan2_deg((float)s_m01[tid], (float)s_m10[tid]);
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        float a, b;
        sincos_det(angle * factorPI, &a, &b);
        s_ang[tid] = angle;
        s_a[tid] = a;
        s_b[tid] = b;
    }
    __syncthreads();
    /* ---- 3. steered BRIEF on the blurred level ---- */
    for (int k = 0; k < SPW; k++) {
        const int sl = wave * SPW + k;
        const int level = __builtin_amdgcn_readfirstlane(s_lvl[sl]);
        if (level < 0) continue;
        const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_key[sl]);
        const int oi = __builtin_amdgcn_readfirstlane(s_oi[sl]);
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_a[sl])));
        const float b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_b[sl])));
        const LevelGeom &L = g.lv[level];
        const int px = KEY_X(key) + ORB_MIN_BORDER, py = KEY_Y(key) + ORB_MIN_BORDER;
        const int pitch = L.pitch;
        /* rotated pattern points stay within +-19 px of the keypoint (|p| <= 13 per axis, EDGE_THRESHOLD 19) */
        const uint8_t *bc = level_ptr(blur, g, frame, level) + (int64_t)(py - 20) * pitch + (px - 20);
        uint64_t words[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int bit = j * 64 + lane;
            const int8_t *p = c_pattern + bit * 4;
            const float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
            const int r0 = (int)rintf(x0 * b + y0 * a), q0 = (int)rintf(x0 * a - y0 * b);
            const int r1 = (int)rintf(x1 * b + y1 * a), q1 = (int)rintf(x1 * a - y1 * b);
            const int t0 = bc[(uint32_t)(__mul24(r0 + 20, pitch) + q0 + 20)], t1 = bc[(uint32_t)(__mul24(r1 + 20, pitch) + q1 + 20)];
            words[j] = __ballot(t0 < t1);
        }
        if (lane < 4) {
            uint64_t w = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
            *(uint64_t *)(desc + ((int64_t)frame * cap + oi) * 32 + lane * 8) = w;
        }
        if (lane == 0) {
            OrbKeyPoint kp;
            float fx = (float)px, fy = (float)py;
            if (level != 0) { /* :1055-1061 */
                fx *= L.scale;
                fy *= L.scale;
            }
            kp.x = fx;
            kp.y = fy;
            kp.size = L.patch_size;
            kp.angle = s_ang[sl];
            kp.response = (float)KEY_R(key);
            kp.octave = level;
            kp.class_id = -1;
            kps[(int64_t)frame * cap + oi] = kp;
        }
    }
}

/* upload of the sampling pattern, ONCE per device (the table never changes), on the caller's stream: called from
   orb_create, never from a launch path.  Not hipMemcpyToSymbol: that one runs on the legacy stream, which ROCm 7.2
   refuses while another host thread's extractor captures its launch graph. */
hipError_t orb_upload_constants(int device, hipStream_t stream) {
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (device >= 0 && device < 64 && done[device]) return hipSuccess;
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_pattern), h_pattern, sizeof(h_pattern), 0, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess && device >= 0 && device < 64) done[device] = true;
    return e;
}

hipError_t launch_describe(const OrbGeom &g, const OrbBuffers &b, int n_frames, OrbKeyPoint *d_kps, uint8_t *d_desc,
                           int cap, int32_t *d_counts, hipStream_t st) {
    static const int wide_min = getenv("ORBSLAM_AMD_DESC_BATCH_MIN") ? atoi(getenv("ORBSLAM_AMD_DESC_BATCH_MIN")) : 16;
    /* Slots per wavefront of a batch launch.  More slots share one issue of the trig phase (VALU instructions per 256
       frames: 72 M at 2, 65 M at 4, 55 M at 8, 53 M at 16; 111 M for the wavefront-per-keypoint kernel), but fewer, longer
       workgroups per frame put more FRAMES in flight on an XCD at once -- 256 resident workgroups are 15 frames at 16
       slots, 4 at 4 -- and a frame's pyramids are 2 MB of a 4 MB L2: HBM-side fetch per launch 0.23 / 0.24 / 0.40 /
       0.60 GB.  Throughput is the same within noise (156-157 k frames/s), so the default keeps the traffic down. */
    static const int spw = getenv("ORBSLAM_AMD_DESC_SPW") ? atoi(getenv("ORBSLAM_AMD_DESC_SPW")) : 4;
#define DESC_LAUNCH(SPW)                                                                                               \
    {                                                                                                                  \
        dim3 grid((g.total_out_cap + 4 * SPW - 1) / (4 * SPW), n_frames);                                              \
        hipLaunchKernelGGL(k_describe<SPW>, grid, dim3(256), 0, st, g, (const uint8_t *)b.pyr, (const uint8_t *)b.blur, \
                           (const uint32_t *)b.lvl_kp, (const int32_t *)b.lvl_cnt, d_kps, d_desc, cap, d_counts);      \
    }
    if (n_frames < wide_min) DESC_LAUNCH(1)
    else if (spw >= 16) DESC_LAUNCH(16)
    else if (spw >= 8) DESC_LAUNCH(8)
    else if (spw >= 4) DESC_LAUNCH(4)
    else DESC_LAUNCH(2)
    return hipGetLastError();
}
