/* A plain C host of libyagi_hip.so (no Python, no torch): the headline stream and one FirFilter block through the
 * C ABI exactly as INTEGRATION.md section 2 shows.  Prints values that tests/test_gpu_chost.py compares with the
 * Python mirror's results on the same generated input.
 *   cc capi_host.c -I../../include -L../../yagi_amd -lyagi_hip -Wl,-rpath,<repo>/yagi_amd -lm -o capi_host */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <math.h>
#include "yagi_hip.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != YAGI_OK) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, yagi_hip_last_error());      \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(void) {
    enum { TAPS = 256, NFFT = 4096, NFRAMES = 64 };
    const size_t n = (size_t)NFFT * NFRAMES;
    int ndev = 0;
    CHECK(yagi_hip_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no GPU\n"); return 2; }
    float h[TAPS];
    CHECK(yagi_hip_fir_design_kaiser(TAPS, 0.2f, 60.0f, 0.0f, h));

    yagi_cf32 *dx = NULL, *dy = NULL;
    CHECK(yagi_hip_malloc((void **)&dx, n * sizeof(yagi_cf32)));
    CHECK(yagi_hip_malloc((void **)&dy, n * sizeof(yagi_cf32)));
    CHECK(yagi_hip_gen_complex_dev(0x59414749ull + 2, 0, n, dx, NULL));

    /* firfilt_crcf -> 4096-point FFT stream, two calls (state carried) */
    yagi_hip_firfft_crcf s;
    CHECK(yagi_hip_firfft_crcf_create(h, TAPS, NFFT, &s));
    CHECK(yagi_hip_firfft_crcf_set_scale(s, 0.4f));
    CHECK(yagi_hip_firfft_crcf_execute_dev(s, dx, NFRAMES / 2, dy));
    CHECK(yagi_hip_firfft_crcf_execute_dev(s, dx + (n / 2), NFRAMES / 2, dy + (n / 2)));
    CHECK(yagi_hip_device_synchronize());
    yagi_cf32 *spec = (yagi_cf32 *)malloc(n * sizeof(yagi_cf32));
    CHECK(yagi_hip_memcpy_d2h(spec, dy, n * sizeof(yagi_cf32)));
    double e = 0.0;
    for (size_t i = 0; i < n; i++) e += (double)spec[i].re * spec[i].re + (double)spec[i].im * spec[i].im;
    printf("stream_energy %.9e\n", e);
    printf("stream_bin %.9e %.9e\n", (double)spec[33 * NFFT + 100].re, (double)spec[33 * NFFT + 100].im);
    CHECK(yagi_hip_firfft_crcf_destroy(s));

    /* FirFilter<Complex32,f32>::execute_block on the same device buffer */
    yagi_hip_firfilt_crcf q;
    CHECK(yagi_hip_firfilt_crcf_create(h, TAPS, &q));
    CHECK(yagi_hip_firfilt_crcf_set_scale(q, 0.4f));
    CHECK(yagi_hip_firfilt_crcf_execute_block_dev(q, dx, n, dy));
    CHECK(yagi_hip_device_synchronize());
    CHECK(yagi_hip_memcpy_d2h(spec, dy, 8 * sizeof(yagi_cf32)));
    printf("fir_y7 %.9e %.9e\n", (double)spec[7].re, (double)spec[7].im);
    size_t len = 0;
    CHECK(yagi_hip_firfilt_crcf_get_length(q, &len));
    printf("fir_len %zu\n", len);
    /* per-sample calls after a block call: served from the host mirror of the window (one download, then no launches) */
    {
        enum { NPS = 200000 };
        yagi_cf32 xs = {0.25f, -0.5f}, ys = {0.0f, 0.0f}, acc = {0.0f, 0.0f};
        struct timespec t0, t1;
        CHECK(yagi_hip_firfilt_crcf_execute_one(q, xs, &ys));          /* pays the one device -> host window copy */
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < NPS; i++) {
            xs.re = (float)(i & 7) * 0.125f;
            CHECK(yagi_hip_firfilt_crcf_execute_one(q, xs, &ys));
            acc.re += ys.re;
            acc.im += ys.im;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("per_sample_ns %.1f\n", ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / NPS);
        printf("per_sample_acc %.6e %.6e\n", (double)acc.re, (double)acc.im);
    }
    CHECK(yagi_hip_firfilt_crcf_destroy(q));

    /* error path: message available through yagi_hip_last_error() */
    yagi_hip_fft bad = NULL;
    int rc = yagi_hip_fft_create(0, YAGI_FFT_FORWARD, &bad);
    printf("fft0_status %d\n", rc);
    free(spec);
    CHECK(yagi_hip_free(dx));
    CHECK(yagi_hip_free(dy));
    return rc == YAGI_ERR_CONFIG ? 0 : 3;
}
