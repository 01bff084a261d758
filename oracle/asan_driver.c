/* asan_driver.c -- the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle asan): every entry point on
 * exact-size heap buffers of awkward geometry -- every format, 1..8 channels, rates whose hop is not a whole number of
 * samples, lengths around the block / window / resampler-tap limits (0, 1, 32, 33, one hop, four hops +- 1, ...).  A read
 * or write past a buffer, a signed overflow or a bad shift aborts the program; the numbers themselves are the business
 * of tests/test_oracle_kat.py.  Test infrastructure only. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hr_oracle.h"

static int fmt_bytes(int fmt) { return fmt == HRO_FMT_S16 ? 2 : fmt == HRO_FMT_S24 ? 3 : fmt == HRO_FMT_F64 ? 8 : 4; }

int main(void)
{
    static const uint32_t rates[] = {8000, 11025, 22050, 44100, 48000, 96000, 192000};
    long calls = 0;
    for (size_t ri = 0; ri < sizeof rates / sizeof rates[0]; ri++) {
        const uint32_t fs = rates[ri], hop = hro_hop(fs);
        const uint64_t lens[] = {0, 1, 32, 33, hop - 1, hop, 4ull * hop - 1, 4ull * hop + 17, 7ull * hop + 5};
        for (size_t li = 0; li < sizeof lens / sizeof lens[0]; li++) {
            const uint64_t frames = lens[li];
            if (fs >= 96000 && frames > 5ull * hop) continue; /* (keeps the run short) */
            for (int ch = 1; ch <= HRO_MAX_CH; ch = (ch == 1 ? 2 : ch == 2 ? 6 : 9)) {
                float *f32 = malloc(frames * ch * sizeof(float) + 1);
                hro_synth_track(fs + ch, fs, ch, frames, f32);
                for (int fmt = 0; fmt <= HRO_FMT_F64; fmt++) {
                    const size_t bytes = (size_t)frames * ch * fmt_bytes(fmt);
                    unsigned char *buf = malloc(bytes ? bytes : 1); /* exact size: one byte too far is a report */
                    for (uint64_t i = 0; i < frames * ch; i++) {
                        const double x = f32[i];
                        if (fmt == HRO_FMT_F32) ((float *)buf)[i] = (float)x;
                        else if (fmt == HRO_FMT_F64) ((double *)buf)[i] = x;
                        else if (fmt == HRO_FMT_S16) ((int16_t *)buf)[i] = (int16_t)lrint(fmax(-32768.0, fmin(32767.0, x * 32768.0)));
                        else if (fmt == HRO_FMT_S32) ((int32_t *)buf)[i] = (int32_t)llrint(fmax(-2147483648.0, fmin(2147483647.0, x * 2147483648.0)));
                        else {
                            const int32_t q = (int32_t)lrint(fmax(-8388608.0, fmin(8388607.0, x * 8388608.0)));
                            buf[3 * i] = q & 0xff; buf[3 * i + 1] = (q >> 8) & 0xff; buf[3 * i + 2] = (q >> 16) & 0xff;
                        }
                    }
                    hro_track t = {buf, frames, fs, (uint16_t)ch, (uint16_t)fmt, NULL};
                    hro_result r;
                    /* (one gate / true-peak combination per buffer: over the formats and channel counts every pair comes up) */
                    if (hro_measure(&t, (fmt + (int)li) % 3, (fmt + ch) % 3, &r) != 0) { fprintf(stderr, "hro_measure failed\n"); return 2; }
                    calls++;
                    if (fs <= 48000 && frames <= 5ull * hop && fmt < 2) { hro_loudnorm_measure(&t, &r); calls++; }
                    uint8_t *out = malloc(frames * ch * 3 + 1);
                    hro_apply_gain(&t, 1.5, fmt & 1, out);
                    free(out);
                    free(buf);
                }
                free(f32);
            }
        }
    }
    printf("oracle under asan + ubsan: %ld measurements, no report\n", calls);
    return 0;
}
