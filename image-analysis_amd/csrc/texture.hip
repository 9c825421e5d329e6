// MPEG-7 homogeneous texture per segment (Radon transform + Gabor channel energies).
// Replaces MPEG7Descriptors.radon_feature_extract / mpeg7_homogeneus_texture
// (scripts/mpeg7Descriptors.py:277-365, const_data.py).
//
// Per segment (value image: the grey level on the segment and on pixels outside every segment,
// 0 on other segments -- segment_generator :28-38 -- as float32(g)/255 padded to N x N,
// N = max(H, W), :311-319):
//   1. scipy.ndimage.rotate(image, 180*per/329 deg, reshape=False, order=3, mode='constant'),
//      per = 0..328, summed over axis 0 (:323-331).  scipy's algorithm is restated: cubic
//      B-spline prefilter with mirror boundaries along both axes (pole sqrt(3)-2, exact causal
//      initialisation), then for every output pixel the input point
//      (i*m00 + j*m01) + off0, (i*m10 + j*m11) + off1; outside [0, N-1] the value is 0, inside
//      the 4 x 4 weighted sum of coefficients with mirrored indices.
//   2. fftshift(fft(ifftshift(sinogram, axes=1)), axes=1) (:334-341) as a direct DFT (N is a few
//      hundred: 329 N^2 complex multiply-adds).
//   3. for the 30 Gabor channels (:287-305, :349-359): z = (g * radius) * F, p = z^2 as a
//      COMPLEX square (the reference does not take the modulus), sum_p = sum p,
//      sum_q = sum (p - sum_p)^2.  The host turns the two complex sums into
//      e = |log10(1 + sum_p)|, d = |log10(1 + sqrt(sum_q))|.
//   4. mean and population standard deviation of the padded image (:361-364).
//
// Everything is float64 as in scipy / numpy; results differ from them in summation order only.
// All reductions are done by one workgroup per (segment, quantity) in a fixed order, so the
// output is reproducible run to run.
#include "common.h"
#include "../../include/imgan.h"
#include <math.h>

namespace {

constexpr int NANG = 329;      // c_num (:310)
constexpr int NCH = 30;        // gabor_radial * gabor_angular (:13-14)

struct TexWs {
    double* coef;      // [chunk][N][N]
    double* sino;      // [chunk][NANG][N]
    double* fre;       // [chunk][NANG][N]
    double* fim;       // [chunk][NANG][N]
};

__device__ __forceinline__ double block_sum(double v, double* red) {
    const int t = threadIdx.x, nt = blockDim.x;
    red[t] = v;
    __syncthreads();
    for (int off = nt >> 1; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    const double out = red[0];
    __syncthreads();
    return out;
}

// value image + mean / std.  One workgroup per segment of the chunk.
__global__ __launch_bounds__(1024) void tex_fill_kernel(const uint8_t* __restrict__ grey,
                                                        const int32_t* __restrict__ labels, int H,
                                                        int W, int N, int seg0, TexWs ws,
                                                        double* __restrict__ out, int out_stride) {
    __shared__ double red[1024];
    const int c = blockIdx.x, s = seg0 + c;
    double* img = ws.coef + (size_t)c * N * N;
    double sum = 0.0;
    for (int p = threadIdx.x; p < N * N; p += 1024) {
        const int i = p / N, j = p - i * N;
        double v = 0.0;
        if (i < H && j < W) {
            const int l = labels[(size_t)i * W + j];
            if (l == s || l < 0) v = (double)((float)grey[(size_t)i * W + j] / 255.0f);   // :311
        }
        img[p] = v;
        sum += v;
    }
    const double mean = block_sum(sum, red) / ((double)N * N);
    double dev = 0.0;
    for (int p = threadIdx.x; p < N * N; p += 1024) {
        const double d = img[p] - mean;
        dev += d * d;
    }
    const double var = block_sum(dev, red) / ((double)N * N);
    if (threadIdx.x == 0) {
        out[(size_t)s * out_stride + 4 * NCH] = mean;
        out[(size_t)s * out_stride + 4 * NCH + 1] = sqrt(var);
    }
}

// cubic B-spline prefilter of one line (scipy ni_splines.c, mirror boundaries)
__device__ void spline_line(double* __restrict__ c, int n, size_t stride) {
    if (n < 2) return;
    const double z = sqrt(3.0) - 2.0;
    for (int i = 0; i < n; ++i) c[i * stride] *= 6.0;            // gain (1 - z)(1 - 1/z)
    const double zn1 = pow(z, (double)(n - 1));
    double c0 = c[0] + zn1 * c[(size_t)(n - 1) * stride];
    double zi = z;
    for (int i = 1; i < n - 1; ++i) {
        c0 += zi * (c[i * stride] + zn1 * c[(size_t)(n - 1 - i) * stride]);
        zi *= z;
    }
    c[0] = c0 / (1.0 - zn1 * zn1);
    for (int i = 1; i < n; ++i) c[i * stride] += z * c[(size_t)(i - 1) * stride];
    c[(size_t)(n - 1) * stride] =
        (z * c[(size_t)(n - 2) * stride] + c[(size_t)(n - 1) * stride]) * z / (z * z - 1.0);
    for (int i = n - 2; i >= 0; --i) c[i * stride] = z * (c[(size_t)(i + 1) * stride] - c[i * stride]);
}

// axis 0 first (lines = columns), then axis 1 (lines = rows), as scipy.ndimage.spline_filter
__global__ __launch_bounds__(64) void tex_prefilter_kernel(TexWs ws, int N, int axis) {
    const int line = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y;
    if (line >= N) return;
    double* img = ws.coef + (size_t)c * N * N;
    if (axis == 0)
        spline_line(img + line, N, (size_t)N);
    else
        spline_line(img + (size_t)line * N, N, 1);
}

__device__ __forceinline__ int mirror_index(int idx, int n) {
    if (n <= 1) return 0;
    const int s2 = 2 * n - 2;
    if (idx < 0) {
        idx = s2 * (-idx / s2) + idx;
        idx = idx <= 1 - n ? idx + s2 : -idx;
    } else if (idx >= n) {
        idx -= s2 * (idx / s2);
        if (idx >= n) idx = s2 - idx;
    }
    return idx;
}

__device__ __forceinline__ void spline_weights(double x, double* w) {
    const double y = x - floor(x), z = 1.0 - y;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = z * z * z / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
}

// sinogram[per][j] = sum_i rotate(image, angle_per)[i][j].  Grid (NANG, chunk); a thread owns
// output column j and walks down the rows.  rot: double [NANG][6] = m00, m01, m10, m11, off0, off1.
__global__ __launch_bounds__(256) void tex_radon_kernel(TexWs ws, int N,
                                                        const double* __restrict__ rot) {
    const int per = blockIdx.x, c = blockIdx.y;
    const double* co = ws.coef + (size_t)c * N * N;
    const double* r = rot + (size_t)per * 6;
    const double m00 = r[0], m01 = r[1], m10 = r[2], m11 = r[3], o0 = r[4], o1 = r[5];
    const double top = (double)(N - 1);
    for (int j = threadIdx.x; j < N; j += 256) {
        double acc = 0.0;
        for (int i = 0; i < N; ++i) {
            const double c0 = ((double)i * m00 + (double)j * m01) + o0;
            const double c1 = ((double)i * m10 + (double)j * m11) + o1;
            if (c0 < 0.0 || c0 > top || c1 < 0.0 || c1 > top) continue;     // mode='constant'
            const int s0 = (int)floor(c0) - 1, s1 = (int)floor(c1) - 1;
            double w0[4], w1[4];
            spline_weights(c0, w0);
            spline_weights(c1, w1);
            double t = 0.0;
            if (s0 >= 0 && s0 + 3 < N && s1 >= 0 && s1 + 3 < N) {
                const double* p = co + (size_t)s0 * N + s1;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) t += (p[(size_t)a * N + b] * w0[a]) * w1[b];
            } else {
                for (int a = 0; a < 4; ++a) {
                    const double* row = co + (size_t)mirror_index(s0 + a, N) * N;
                    for (int b = 0; b < 4; ++b) t += (row[mirror_index(s1 + b, N)] * w0[a]) * w1[b];
                }
            }
            acc += t;
        }
        ws.sino[((size_t)c * NANG + per) * N + j] = acc;
    }
}

// F[per][k] = fftshift(fft(ifftshift(sino[per])))[k].  Grid (NANG, chunk), a thread per k.
__global__ __launch_bounds__(256) void tex_dft_kernel(TexWs ws, int N) {
    const int per = blockIdx.x, c = blockIdx.y;
    const double* x = ws.sino + ((size_t)c * NANG + per) * N;
    const int h = N / 2;
    for (int k = threadIdx.x; k < N; k += 256) {
        const int m = ((k - h) % N + N) % N;                     // fftshift: F'[k] = X[(k - N/2) mod N]
        double re = 0.0, im = 0.0;
        for (int n = 0; n < N; ++n) {
            const double v = x[(n + h) % N];                     // ifftshift: x'[n] = x[(n + N/2) mod N]
            const long long t = ((long long)m * n) % N;
            double sn, cs;
            sincospi(2.0 * (double)t / (double)N, &sn, &cs);
            re += v * cs;
            im -= v * sn;
        }
        ws.fre[((size_t)c * NANG + per) * N + k] = re;
        ws.fim[((size_t)c * NANG + per) * N + k] = im;
    }
}

// chan: double [NCH][4] = centre frequency, 2 rad_std^2, centre angle (rad), 2 ang_std^2.
// PASS 0: sum p; PASS 1: sum (p - sum_p)^2 with sum_p read back from `out`.
// out: double [nseg][out_stride]: [ch*4 + 0..1] = sum_p (re, im), [ch*4 + 2..3] = sum_q.
template <int PASS>
__global__ __launch_bounds__(1024) void tex_gabor_kernel(TexWs ws, int N,
                                                         const double* __restrict__ chan,
                                                         const double* __restrict__ angle_p,
                                                         const double* __restrict__ radius_p,
                                                         int seg0, double* __restrict__ out,
                                                         int out_stride) {
    __shared__ double red[1024];
    const int ch = blockIdx.x, c = blockIdx.y;
    const double f0 = chan[ch * 4], rv = chan[ch * 4 + 1], a0 = chan[ch * 4 + 2], av = chan[ch * 4 + 3];
    double* o = out + (size_t)(seg0 + c) * out_stride + ch * 4;
    const double sr = PASS ? o[0] : 0.0, si = PASS ? o[1] : 0.0;
    const double* fre = ws.fre + (size_t)c * NANG * N;
    const double* fim = ws.fim + (size_t)c * NANG * N;
    double ar = 0.0, ai = 0.0;
    for (int p = threadIdx.x; p < NANG * N; p += 1024) {
        const int per = p / N, k = p - per * N;
        const double rad = radius_p[k], ang = angle_p[per];
        const double dr = (rad - 0.5) - f0, da = ang - a0;                       // :301-302
        const double g = exp(-(dr * dr) / rv) * exp(-(da * da) / av);
        const double gr = g * rad;                                               // :351
        const double zr = gr * fre[p], zi = gr * fim[p];
        double pr = zr * zr - zi * zi, pi = 2.0 * zr * zi;                       // complex square
        if (PASS) {
            const double qr = pr - sr, qi = pi - si;                             // :354
            pr = qr * qr - qi * qi;
            pi = 2.0 * qr * qi;
        }
        ar += pr;
        ai += pi;
    }
    const double tr = block_sum(ar, red), ti = block_sum(ai, red);
    if (threadIdx.x == 0) {
        o[PASS * 2] = tr;
        o[PASS * 2 + 1] = ti;
    }
}

size_t per_seg_bytes(int N) { return ((size_t)N * N + 3 * (size_t)NANG * N) * 8; }
int chunk_of(int nseg) { return nseg < 8 ? nseg : 8; }

}  // namespace

extern "C" size_t imgan_texture_scratch_bytes(int H, int W, int nseg) {
    if (H < 1 || W < 1 || nseg < 1) return 0;
    const int N = H > W ? H : W;
    return imgan::align_up((size_t)chunk_of(nseg) * per_seg_bytes(N), 256);
}

extern "C" int imgan_homogeneous_texture(const uint8_t* grey, const int32_t* labels, int H, int W,
                                         int nseg, const double* rot, const double* chan,
                                         const double* angle_p, const double* radius_p, double* out,
                                         void* scratch, imgan_stream_t stream) {
    IMGAN_REQUIRE(grey && labels && rot && chan && angle_p && radius_p && out && scratch,
                  "imgan_homogeneous_texture: null pointer");
    IMGAN_REQUIRE(H >= 1 && W >= 1 && nseg >= 1, "imgan_homogeneous_texture: bad shape");
    const int N = H > W ? H : W;
    IMGAN_REQUIRE(N >= 4 && N <= 16384, "imgan_homogeneous_texture: side must be within 4..16384");
    hipStream_t st = (hipStream_t)stream;
    const int chunk = chunk_of(nseg);
    const int stride = 4 * NCH + 2;
    TexWs ws;
    char* p = (char*)scratch;
    ws.coef = (double*)p;
    ws.sino = ws.coef + (size_t)chunk * N * N;
    ws.fre = ws.sino + (size_t)chunk * NANG * N;
    ws.fim = ws.fre + (size_t)chunk * NANG * N;
    for (int s0 = 0; s0 < nseg; s0 += chunk) {
        const int nc = nseg - s0 < chunk ? nseg - s0 : chunk;
        hipLaunchKernelGGL(tex_fill_kernel, dim3(nc), dim3(1024), 0, st, grey, labels, H, W, N, s0, ws,
                           out, stride);
        dim3 gl((unsigned)imgan::cdiv(N, 64), nc);
        hipLaunchKernelGGL(tex_prefilter_kernel, gl, dim3(64), 0, st, ws, N, 0);
        hipLaunchKernelGGL(tex_prefilter_kernel, gl, dim3(64), 0, st, ws, N, 1);
        hipLaunchKernelGGL(tex_radon_kernel, dim3(NANG, nc), dim3(256), 0, st, ws, N, rot);
        hipLaunchKernelGGL(tex_dft_kernel, dim3(NANG, nc), dim3(256), 0, st, ws, N);
        hipLaunchKernelGGL(tex_gabor_kernel<0>, dim3(NCH, nc), dim3(1024), 0, st, ws, N, chan, angle_p,
                           radius_p, s0, out, stride);
        hipLaunchKernelGGL(tex_gabor_kernel<1>, dim3(NCH, nc), dim3(1024), 0, st, ws, N, chan, angle_p,
                           radius_p, s0, out, stride);
    }
    return imgan::check_launch("imgan_homogeneous_texture");
}
