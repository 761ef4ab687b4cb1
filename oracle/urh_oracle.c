/*
 * urh_oracle.c -- CPU restatement of the reference's IQ->bits hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (urh_amd/) never imports, links or calls anything under oracle/.
 *
 * Every function follows the reference file:line it cites, statement by statement, in plain
 * single-threaded C.  Build: gcc -O2 -ffp-contract=off -fPIC -shared (see urh_oracle.build() in oracle/urh_oracle.py);
 * -ffp-contract=off because the reference's x86-64 build contains no FMA instructions
 * (SURVEY.md appendix).  libm functions are the host glibc ones, exactly as in the reference
 * (atan2f / sqrtf / sinf / cosf: the Cython module is compiled as C++, so `atan2(float,float)` and
 * `sqrt(float)` bind to the float overloads).
 *
 * Parity pin: tests/test_oracle.py checks every function here against the real reference
 * build (oracle/_ref, built by oracle/build_ref.py from /root/reference) and against the golden
 * vectors under tests/golden/ (made by tests/golden/make_golden.py from the reference's own
 * Python + Cython code and its tests' known answers).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* dtype codes shared with include/urhgpu.h */
enum { DT_I8 = 0, DT_U8 = 1, DT_I16 = 2, DT_U16 = 3, DT_F32 = 4 };
/* modulation codes shared with include/urhgpu.h */
enum { MOD_ASK = 0, MOD_FSK = 1, MOD_PSK = 2, MOD_OTHER = 3 /* e.g. "QAM", "OQPSK": no demod branch */,
       MOD_OQPSK = 4 /* orc_modulate only */ };

#define PAUSE_STATE (-1)

/* signal_functions.pyx:31-44 get_noise_for_mod_type.  "OQPSK" shares -4.0 with PSK; "QAM" gives
 * NOISE_ASK*NOISE_FSK_PSK = -0.0; anything else 0.  Callers pass the sentinel explicitly for
 * MOD_OTHER via orc_noise_for(). */
static float noise_for(int mod) {
    if (mod == MOD_ASK) return 0.0f;
    if (mod == MOD_FSK || mod == MOD_PSK) return -4.0f;
    return 0.0f;
}

static inline float iq_load(const void *p, int dt, int64_t idx) {
    switch (dt) {
        case DT_I8: return (float)((const int8_t *)p)[idx];
        case DT_U8: return (float)((const uint8_t *)p)[idx];
        case DT_I16: return (float)((const int16_t *)p)[idx];
        case DT_U16: return (float)((const uint16_t *)p)[idx];
        default: return ((const float *)p)[idx];
    }
}

/* util.pyx:128-136 get_magnitudes.  Product/sum in the element type: float for float32, C `int`
 * (after integer promotion) for the integer dtypes; sqrt binds to sqrtf for float input and to
 * double sqrt for integer input (C++ overloads). */
void orc_get_magnitudes(const void *iq, int dt, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) {
        if (dt == DT_F32) {
            float re = ((const float *)iq)[2 * i], im = ((const float *)iq)[2 * i + 1];
            out[i] = (double)sqrtf(re * re + im * im);
        } else {
            int re, im;
            switch (dt) {
                case DT_I8: re = ((const int8_t *)iq)[2 * i]; im = ((const int8_t *)iq)[2 * i + 1]; break;
                case DT_U8: re = ((const uint8_t *)iq)[2 * i]; im = ((const uint8_t *)iq)[2 * i + 1]; break;
                case DT_I16: re = ((const int16_t *)iq)[2 * i]; im = ((const int16_t *)iq)[2 * i + 1]; break;
                default: re = ((const uint16_t *)iq)[2 * i]; im = ((const uint16_t *)iq)[2 * i + 1]; break;
            }
            /* int arithmetic wraps for uint16 65535^2*2 exactly as the reference's C `int` does
             * (unsigned wrap used here to avoid UB; same two's-complement bits). */
            int s = (int)((unsigned)(re * re) + (unsigned)(im * im));
            out[i] = sqrt((double)s);
        }
    }
}

/* signal_functions.pyx:246-250 */
static inline float clampf(float x) {
    if (x < -1.0f) x = -1.0f;
    else if (x > 1.0f) x = 1.0f;
    return x;
}

/* signal_functions.pyx:252-330 costa_demod.  result[0] is never written by the reference
 * (np.empty); we leave out[0] untouched too (callers compare from index 1). */
static int costa_demod(const void *iq, int dt, int64_t n, float noise_sqrd, int loop_order,
                       float bandwidth, float *out) {
    float damping = (float)(sqrt(2.0) / 2.0);             /* default arg `float damping=sqrt(2.0)/2.0` */
    /* :253-254 as Cython emits them: (double)((4.0*damping)*bandwidth) / ((1.0 + ((2.0*damping)*bandwidth)) +
     * (bandwidth*bandwidth)) -- note bandwidth*bandwidth is a FLOAT product -- stored to float. */
    double den = (1.0 + ((2.0 * (double)damping) * (double)bandwidth)) + (double)(bandwidth * bandwidth);
    float alpha = (float)(((4.0 * (double)damping) * (double)bandwidth) / den);
    float beta = (float)(((4.0 * (double)bandwidth) * (double)bandwidth) / den);
    float scale, shift;
    switch (dt) {                                          /* :267-283 */
        case DT_I8: scale = 127.5f; shift = 0.5f; break;
        case DT_U8: scale = 127.5f; shift = -127.5f; break;
        case DT_I16: scale = 32767.5f; shift = 0.5f; break;
        case DT_U16: scale = 65535.0f; shift = -32767.5f; break;
        case DT_F32: scale = 1.0f; shift = 0.0f; break;
        default: return -1;
    }
    if (loop_order > 4) loop_order = 4;                    /* :285-287 */
    float costa_freq = 0, costa_error = 0, costa_phase = 1.5f;
    const float _Complex imag_unit = 1.0f * I;
    for (int64_t i = 1; i < n; i++) {
        float real = iq_load(iq, dt, 2 * i), imag = iq_load(iq, dt, 2 * i + 1);
        if (real * real + imag * imag <= noise_sqrd) { out[i] = -4.0f; continue; }
        float real_float = (real + shift) / scale;
        float imag_float = (imag + shift) / scale;
        float _Complex current_sample = real_float + imag_unit * imag_float;
        float _Complex nco_out = cosf(-costa_phase) + imag_unit * sinf(-costa_phase);
        float _Complex z = nco_out * current_sample;
        if (loop_order == 2) {
            costa_error = cimagf(z) * crealf(z);
        } else if (loop_order == 4) {
            float f1 = crealf(z) > 0.0f ? 1.0f : -1.0f;
            float f2 = cimagf(z) > 0.0f ? 1.0f : -1.0f;
            costa_error = f1 * cimagf(z) - f2 * crealf(z);
        }
        costa_error = clampf(costa_error);
        costa_freq += beta * costa_error;
        costa_phase += costa_freq + alpha * costa_error;
        while (costa_phase > (2 * M_PI)) costa_phase -= 2 * M_PI;    /* double compare / subtract */
        while (costa_phase < (-2 * M_PI)) costa_phase += 2 * M_PI;
        costa_freq = clampf(costa_freq);
        if (loop_order == 2) out[i] = crealf(z);
        else if (loop_order == 4) out[i] = (float)((2.0 * (double)crealf(z)) + (double)cimagf(z));   /* double literal 2.0 */
    }
    return 0;
}

/* signal_functions.pyx:333-378 afp_demod.  `out` has n floats.  Returns 0, or -1 for a bad dtype.
 * For mod==MOD_PSK out[0] is left untouched when n > 2 (reference: np.empty). */
int orc_afp_demod(const void *iq, int dt, int64_t n, float noise_mag, int mod, int mod_order,
                  float costas_bw, float noise_sentinel_other, float *out) {
    if (n <= 2) { for (int64_t i = 0; i < n; i++) out[i] = 0.0f; return 0; }   /* :335-336 */
    float NOISE = (mod == MOD_OTHER) ? noise_sentinel_other : noise_for(mod);
    float noise_sqrd = noise_mag * noise_mag;
    float max_magnitude;
    switch (dt) {                                           /* :343-354: double sqrt of int consts -> float */
        case DT_I8: max_magnitude = (float)sqrt((double)(127 * 127 + 128 * 128)); break;
        case DT_U8: max_magnitude = (float)sqrt((double)(255 * 255)); break;
        case DT_I16: max_magnitude = (float)sqrt(2147418113.0); break;                                /* 32768^2+32767^2 */
        case DT_U16: max_magnitude = (float)sqrt((double)4294836225.0); break;                      /* 65535*65535 */
        case DT_F32: max_magnitude = (float)sqrt(2.0); break;
        default: return -1;
    }
    if (mod == MOD_PSK) return costa_demod(iq, dt, n, noise_sqrd, mod_order, costas_bw, out);
    const float _Complex imag_unit = 1.0f * I;
    for (int64_t i = 0; i < n; i++) out[i] = 0.0f;          /* np.zeros */
    out[0] = NOISE;                                         /* :361 */
    for (int64_t i = 1; i < n; i++) {
        float real = iq_load(iq, dt, 2 * i), imag = iq_load(iq, dt, 2 * i + 1);
        float magnitude = real * real + imag * imag;
        if (magnitude <= noise_sqrd) { out[i] = NOISE; continue; }
        if (mod == MOD_ASK) {
            out[i] = (float)((double)sqrtf(magnitude) / (double)max_magnitude);   /* :372 */
        } else if (mod == MOD_FSK) {
            /* :375  (s[i-1,0] - 1j*s[i-1,1]) * (real + 1j*imag) with float-complex operands;
             * written with C99 _Complex so that signed zeros / NaN recovery (__mulsc3) follow the
             * same compiler-runtime rules as the C++ std::complex<float> code Cython generates. */
            float pr = iq_load(iq, dt, 2 * (i - 1)), pi_ = iq_load(iq, dt, 2 * (i - 1) + 1);
            float _Complex a = CMPLXF(pr, 0.0f) - imag_unit * CMPLXF(pi_, 0.0f);
            float _Complex b = CMPLXF(real, 0.0f) + imag_unit * CMPLXF(imag, 0.0f);
            float _Complex tmp = a * b;
            out[i] = atan2f(cimagf(tmp), crealf(tmp));      /* :376 */
        }
    }
    return 0;
}

/* signal_functions.pyx:380-390 */
void orc_get_center_thresholds(float center, float spacing, int order, float *out) {
    int n = order / 2;
    for (int i = 0; i < n; i++) out[i] = center - (n - (i + 1)) * spacing;
    for (int i = n; i < order - 1; i++) out[i] = center + (i + 1 - n) * spacing;
}

/* signal_functions.pyx:392-495 grab_pulse_lens: literal serial state machine.
 * out_rows must hold 2*n int64 (the reference allocates int64[n,2]).  Returns the row count. */
int64_t orc_grab_pulse_lens(const float *samples, int64_t num_samples, float center, uint16_t tolerance,
                            int mod, uint32_t samples_per_symbol, uint8_t bits_per_symbol,
                            float center_spacing, float noise_sentinel_other, int64_t *result) {
    int is_ask = (mod == MOD_ASK);
    int64_t pulse_length = 0, cur_index = 0, consecutive_pause = 0;
    float s = 0, s_prev = 0;
    int cur_state = 0, new_state = 0, tmp_state = 0;
    float NOISE = (mod == MOD_OTHER) ? noise_sentinel_other : noise_for(mod);
    int modulation_order = 1 << bits_per_symbol;
    if (num_samples == 0) return 0;
    float *thresholds = (float *)malloc(sizeof(float) * (modulation_order > 1 ? modulation_order - 1 : 1));
    orc_get_center_thresholds(center, center_spacing, modulation_order, thresholds);
    int64_t *state_count = (int64_t *)calloc(modulation_order, sizeof(int64_t));

    s_prev = samples[0];
    if (s_prev == NOISE) {
        cur_state = PAUSE_STATE;
    } else {
        cur_state = modulation_order - 1;
        for (int k = 0; k < modulation_order - 1; k++)
            if (s <= thresholds[k]) { cur_state = k; break; }     /* s is still 0.0 here (:427) */
    }
    for (int64_t i = 0; i < num_samples; i++) {
        pulse_length += 1;
        s = samples[i];
        if (s == NOISE) {
            tmp_state = PAUSE_STATE;
        } else {
            tmp_state = modulation_order - 1;
            for (int k = 0; k < modulation_order - 1; k++)
                if (s <= thresholds[k]) { tmp_state = k; break; }
        }
        if (tmp_state == PAUSE_STATE) consecutive_pause += 1; else consecutive_pause = 0;
        for (int j = 0; j < modulation_order; j++) {
            if (j == tmp_state) state_count[j] += 1; else state_count[j] = 0;
        }
        if (cur_state == tmp_state) continue;
        new_state = -42;
        if (consecutive_pause > tolerance) {
            new_state = PAUSE_STATE;
        } else {
            for (int j = 0; j < modulation_order; j++)
                if (state_count[j] > tolerance) { new_state = j; break; }
        }
        if (new_state == -42) continue;
        if (is_ask && cur_state == PAUSE_STATE && (pulse_length - tolerance) < (int64_t)samples_per_symbol)
            cur_state = 0;
        if (cur_index > 0 && result[2 * (cur_index - 1)] == cur_state) {
            result[2 * (cur_index - 1) + 1] += pulse_length - tolerance;
        } else {
            result[2 * cur_index] = cur_state;
            result[2 * cur_index + 1] = pulse_length - tolerance;
            cur_index += 1;
        }
        pulse_length = tolerance;
        cur_state = new_state;
    }
    if (cur_index < num_samples) {                              /* :485-493 */
        if (cur_index > 0 && result[2 * (cur_index - 1)] == cur_state) {
            result[2 * (cur_index - 1) + 1] += pulse_length - tolerance;
        } else {
            result[2 * cur_index] = cur_state;
            result[2 * cur_index + 1] = pulse_length - tolerance;
            cur_index += 1;
        }
    }
    free(thresholds);
    free(state_count);
    return cur_index;
}

/* signal_functions.pyx:513-525 fir_filter: scatter form on a zeroed (N+M-1) buffer, first N kept.
 * x, taps, out are interleaved complex64. */
void orc_fir_filter(const float *x, int64_t N, const float *taps, int64_t M, float *out) {
    float _Complex *o = (float _Complex *)calloc((size_t)(N + M - 1 > 0 ? N + M - 1 : 1), sizeof(float _Complex));
    for (int64_t i = 0; i < N; i++) {
        float _Complex xi = CMPLXF(x[2 * i], x[2 * i + 1]);
        for (int64_t j = 0; j < M; j++) o[i + j] += xi * CMPLXF(taps[2 * j], taps[2 * j + 1]);
    }
    for (int64_t i = 0; i < N; i++) { out[2 * i] = crealf(o[i]); out[2 * i + 1] = cimagf(o[i]); }
    free(o);
}

/* signal_functions.pyx:527-542 iir_filter.  As generated: the product (a[j]+0j)*complex128(x) is a
 * full complex128 multiply, ROUNDED to complex64, then accumulated with a complex64 `+=`.
 * No asserting test in the reference; pinned on the real function run here (oracle/_ref) and on its committed outputs
 * (tests/golden/filter/fir_iir.npz). */
void orc_iir_filter(const double *a, int64_t M, const double *b, int64_t N, const float *sig, int64_t len,
                    float *out) {
    for (int64_t i = 0; i < 2 * len; i++) out[i] = 0.0f;
    int64_t start = M > N + 1 ? M : N + 1;
    for (int64_t n = start; n < len; n++) {
        for (int64_t j = 0; j < M; j++) {
            double _Complex t = CMPLX(a[j], 0.0) * CMPLX((double)sig[2 * (n - j)], (double)sig[2 * (n - j) + 1]);
            out[2 * n] += (float)creal(t); out[2 * n + 1] += (float)cimag(t);
        }
        for (int64_t k = 0; k < N; k++) {
            double _Complex t = CMPLX(b[k], 0.0) * CMPLX((double)out[2 * (n - 1 - k)], (double)out[2 * (n - 1 - k) + 1]);
            out[2 * n] += (float)creal(t); out[2 * n + 1] += (float)cimag(t);
        }
    }
}

/* ProtocolAnalyzer.py:323-414 _ppseq_to_bits, restated with flat outputs:
 *   bits[]        all message bits back to back (1 byte per bit)
 *   msg_off[]     nmsg+1 offsets into bits[]
 *   pauses[]      nmsg
 *   pos[]         all bit_sample_pos arrays back to back (only when write_pos)
 *   pos_off[]     nmsg+1 offsets into pos[]
 * Capacities are the caller's responsibility (cap_bits / cap_pos / cap_msg); returns nmsg or -1
 * on overflow.  Python semantics kept: true division in double, int() truncation,
 * `[0]*k` / range(k) empty for k <= 0, negative lengths allowed. */
int64_t orc_ppseq_to_bits(const int64_t *ppseq, int64_t nrows, int64_t samples_per_symbol, int bits_per_symbol,
                          int write_pos, int64_t pause_threshold,
                          uint8_t *bits, int64_t cap_bits, int64_t *msg_off, int64_t *pauses, int64_t cap_msg,
                          int64_t *pos, int64_t cap_pos, int64_t *pos_off) {
    int64_t nbits = 0, npos = 0, nmsg = 0;
    int64_t msg_bits_start = 0, msg_pos_start = 0;
    int64_t start = 0, total_samples = 0;
    int there_was_data = 0;
    int64_t samples_per_bit = (int64_t)((double)samples_per_symbol / (double)bits_per_symbol);   /* :344 int(a/b) */
    msg_off[0] = 0; pos_off[0] = 0;
    if (nrows > 0 && ppseq[0] == -1) { start = 1; total_samples = ppseq[1]; }
    for (int64_t i = start; i < nrows; i++) {
        int64_t cur_pulse_type = ppseq[2 * i], num_samples = ppseq[2 * i + 1];
        double num_symbols_float = (double)num_samples / (double)samples_per_symbol;
        int64_t num_symbols = (int64_t)num_symbols_float;
        double decimal_place = num_symbols_float - (double)num_symbols;
        if (decimal_place > 0.5) num_symbols += 1;
        int64_t k_bits = num_symbols * bits_per_symbol;
        if (cur_pulse_type == -1) {
            if (num_symbols <= pause_threshold || pause_threshold == 0) {
                if (k_bits > 0) {
                    if (nbits + k_bits > cap_bits) return -1;
                    memset(bits + nbits, 0, (size_t)k_bits);
                    nbits += k_bits;
                    if (write_pos) {
                        if (npos + k_bits > cap_pos) return -1;
                        for (int64_t k = 0; k < k_bits; k++) pos[npos++] = total_samples + k * samples_per_bit;
                    }
                }
            } else if (!there_was_data) {
                nbits = msg_bits_start; npos = msg_pos_start;
            } else {
                if (write_pos) {
                    if (npos + 2 > cap_pos) return -1;
                    pos[npos++] = total_samples;
                    pos[npos++] = total_samples + num_samples;
                }
                if (nmsg + 1 > cap_msg) return -1;
                pauses[nmsg] = num_samples;
                nmsg++;
                msg_off[nmsg] = nbits; pos_off[nmsg] = npos;
                msg_bits_start = nbits; msg_pos_start = npos;
                there_was_data = 0;
            }
        } else {
            if (k_bits > 0) {
                if (nbits + k_bits > cap_bits) return -1;
                for (int64_t sy = 0; sy < num_symbols; sy++)
                    for (int b = 0; b < bits_per_symbol; b++)     /* util.number_to_bits: MSB first, zero padded */
                        bits[nbits++] = (uint8_t)((cur_pulse_type >> (bits_per_symbol - 1 - b)) & 1);
            }
            if (!there_was_data && num_symbols > 0) there_was_data = 1;
            if (write_pos && k_bits > 0) {
                if (npos + k_bits > cap_pos) return -1;
                for (int64_t k = 0; k < k_bits; k++) pos[npos++] = total_samples + k * samples_per_bit;
            }
        }
        total_samples += num_samples;
    }
    if (there_was_data) {
        if (write_pos) {
            if (npos + 1 > cap_pos) return -1;
            pos[npos++] = total_samples;
        }
        if (nmsg + 1 > cap_msg) return -1;
        pauses[nmsg] = (ppseq[2 * (nrows - 1)] == -1) ? ppseq[2 * (nrows - 1) + 1] : 0;
        nmsg++;
        msg_off[nmsg] = nbits; pos_off[nmsg] = npos;
    }
    return nmsg;
}

/* test helper: elementwise host-libm atan2f (what the reference's FSK branch calls) */
void orc_atan2f_array(const float *y, const float *x, int64_t n, float *out) {
    for (int64_t i = 0; i < n; i++) out[i] = atan2f(y[i], x[i]);
}

/* ---- modulate_c  (signal_functions.pyx:56-177; rows of SURVEY.md §8f rank 2) --------------------------------
 * ASK / FSK / PSK.  (GFSK goes through numpy's float32 convolve and OQPSK is marked "does not work correctly"
 * in the reference, :179-180; neither is restated.)  Types as in the generated C++:
 *   t             = (float)(i + start) / sample_rate                                         (:160)
 *   current_arg   = (float)((((2.0*M_PI) * f) * t + phi) + phase_correction)   in double      (:164)
 *   out           = (iq)(a * cosf(current_arg)), (iq)(a * sinf(current_arg))                  (:165-166)
 *   FSK phase corrections (:121-137): serial over the symbols,
 *     t = (float)((s_i*sps + start) - 1) / sample_rate;  pc[s] = (float)fmod(pc[s-1] + ((2.0*M_PI)*(f_prev - f))*t, 2.0*M_PI)
 * out: total_symbols*sps + pause samples (x2), zero where nothing is written.  Returns the sample count, -1 on an
 * unsupported modulation.  dt: DT_I8 / DT_I16 / DT_F32 (get_numpy_dtype, :46-54). */
static uint64_t bits_to_number(const uint8_t *bits, int64_t start, int64_t end) {   /* util.pyx:50-61 */
    uint64_t r = 0;
    for (int64_t i = start; i < end; ++i) r = (r << 1) + bits[i];
    return r;
}
static void store_iq(void *out, int dt, int64_t idx, float v) {
    if (dt == DT_F32) ((float *)out)[idx] = v;
    else if (dt == DT_I8) ((int8_t *)out)[idx] = (int8_t)v;
    else ((int16_t *)out)[idx] = (int16_t)v;
}
int64_t orc_modulate(const uint8_t *bits, int64_t num_bits, uint32_t sps, int mod, const float *parameters,
                     int bits_per_symbol, float carrier_amplitude, float carrier_frequency, float carrier_phase,
                     float sample_rate, uint32_t pause, uint32_t start, int dt, void *out) {
    const int is_oqpsk = (mod == MOD_OQPSK);
    if (mod != MOD_ASK && mod != MOD_FSK && mod != MOD_PSK && !is_oqpsk) return -1;
    if (is_oqpsk && bits_per_symbol != 2) return -2;                           /* assert bits_per_symbol == 2 (:120) */
    const uint32_t total_symbols = (uint32_t)(num_bits / bits_per_symbol);
    const int64_t total_samples = (int64_t)total_symbols * sps + pause;
    const int esz = dt == DT_F32 ? 4 : (dt == DT_I8 ? 1 : 2);
    memset(out, 0, (size_t)total_samples * 2 * esz);
    if (num_bits == 0) return total_samples;
    uint8_t *oq = NULL;
    if (is_oqpsk) {                                                            /* get_oqpsk_bits (:179-194), literally */
        oq = (uint8_t *)calloc((size_t)num_bits + 2, 1);
        oq[0] = bits[0];
        oq[num_bits + 1] = bits[num_bits - 1];
        for (int64_t i = 2; i < num_bits - 2; i += 2) { oq[i] = bits[i]; oq[i + 1] = bits[i - 1]; }
        bits = oq;
    }
    float *pc = NULL;
    if (mod == MOD_FSK && total_symbols > 0) {
        pc = (float *)malloc((size_t)total_symbols * sizeof(float));
        pc[0] = 0.0f;
        for (int64_t s = 1; s < total_symbols; ++s) {
            const float f = parameters[bits_to_number(bits, s * bits_per_symbol, (s + 1) * bits_per_symbol)];
            const float fp = parameters[bits_to_number(bits, (s - 1) * bits_per_symbol, s * bits_per_symbol)];
            if (f != fp) {
                const float t = ((float)(((s * (int64_t)sps) + (int64_t)start) - 1)) / sample_rate;
                pc[s] = (float)fmod(pc[s - 1] + (((2.0 * M_PI) * (fp - f)) * t), 2.0 * M_PI);
            } else pc[s] = pc[s - 1];
        }
    }
    for (int64_t s = 0; s < total_symbols; ++s) {
        const uint64_t index = bits_to_number(bits, s * bits_per_symbol, (s + 1) * bits_per_symbol);
        float a = carrier_amplitude, f = carrier_frequency, phi = carrier_phase, corr = 0;
        if (mod == MOD_ASK) { a = parameters[index]; if (a == 0) continue; }
        else if (mod == MOD_FSK) { f = parameters[index]; corr = pc[s]; }
        else phi = parameters[index];
        for (int64_t i = s * (int64_t)sps; i < (s + 1) * (int64_t)sps; ++i) {
            const float t = ((float)(i + (int64_t)start)) / sample_rate;
            const float arg = (float)(((((2.0 * M_PI) * f) * t) + phi) + corr);
            store_iq(out, dt, 2 * i, a * cosf(arg));
            store_iq(out, dt, 2 * i + 1, a * sinf(arg));
        }
    }
    if (is_oqpsk) {                                                            /* :165-169; negative indices wrap around (Cython default) */
        for (int64_t i = 0; i < (int64_t)sps; ++i)
            if (i < total_samples) store_iq(out, dt, 2 * i + 1, 0.0f);
        for (int64_t i = total_samples - pause - sps; i < total_samples - pause; ++i) {
            const int64_t k = i < 0 ? i + total_samples : i;
            if (k >= 0 && k < total_samples) store_iq(out, dt, 2 * k, 0.0f);
        }
    }
    free(pc);
    free(oq);
    return total_samples;
}

/* signal_functions.pyx:196-228 get_gauss_filtered_freqs_phases + the GFSK branch of __modulate (:118-125, :156-163).
 *   frequencies[i] = parameters[symbol(i)]                       float32, one value per sample
 *   t              = np.arange(start, start + n, dtype=float32) / sample_rate
 *                    numpy fills an arange as buf[0] = start, buf[1] = start + 1 (each rounded to float32), then
 *                    buf[i] = buf[0] + i * (buf[1] - buf[0]) in float32; the division is a float32 division
 *   frequencies    = np.convolve(frequencies, gfir, "same")      (or convolve(gfir, frequencies, "same")[:n] when the
 *                    filter is the longer one).  numpy evaluates every output as a float32 BLAS dot product whose summation
 *                    order depends on the host's sdot kernel: THIS STEP IS NOT BIT-REPRODUCIBLE ACROSS HOSTS in the
 *                    reference.  Restated as the exactly-summed (double) dot product rounded once to float32; compared
 *                    with the real reference under a tolerance (tests/test_modulate.py), the device kernel equals this
 *                    restatement bit for bit.
 *   phases[0] = phi;  phases[i+1] = (float)(((2.0*M_PI) * t[i]) * (double)(f[i] - f[i+1]) + phases[i])   (f[i]-f[i+1] in float32)
 *   sample i: t' = (float)(i + start) / sample_rate;  arg = (float)((2.0*M_PI*f[i])*t' + phases[i] + 0);  a*cosf(arg), a*sinf(arg)
 * gfir (gauss_fir, :230-243) is computed by the caller with numpy, exactly as the reference does.
 * freq_in (optional): the filtered frequencies (n floats), used instead of the convolution above -- e.g. numpy's on this host.
 * freq_out / phase_out (optional): the two columns, n floats each.  Returns the sample count; -3: no whole symbol. */
static float arange_f32(uint32_t start, int64_t i) {
    const float s0 = (float)(double)start, s1 = (float)((double)start + 1.0);
    if (i == 0) return s0;
    if (i == 1) return s1;
    const float delta = s1 - s0;
    const float prod = (float)i * delta;
    return s0 + prod;
}
int64_t orc_modulate_gfsk(const uint8_t *bits, int64_t num_bits, uint32_t sps, const float *parameters, int bits_per_symbol,
                          float carrier_amplitude, float carrier_phase, float sample_rate, uint32_t pause, uint32_t start,
                          const float *gfir, int64_t n_taps, const float *freq_in, int dt, void *out, float *freq_out,
                          float *phase_out) {
    const uint32_t total_symbols = (uint32_t)(num_bits / bits_per_symbol);
    const int64_t total_samples = (int64_t)total_symbols * sps + pause;
    const int esz = dt == DT_F32 ? 4 : (dt == DT_I8 ? 1 : 2);
    memset(out, 0, (size_t)total_samples * 2 * esz);
    if (num_bits == 0) return total_samples;
    if (total_symbols == 0) return -3;                               /* len(bits) // num_symbols: ZeroDivisionError (:201) */
    const int bps = (int)(num_bits / total_symbols);                 /* :201 (re-derived; equals bits_per_symbol when it divides) */
    const int64_t n = (int64_t)total_symbols * sps, m = n_taps;
    float *raw = (float *)malloc(sizeof(float) * (size_t)n), *f = (float *)malloc(sizeof(float) * (size_t)n);
    float *ph = (float *)malloc(sizeof(float) * (size_t)n);
    for (int64_t s = 0; s < total_symbols; ++s) {
        const float v = parameters[bits_to_number(bits, s * bps, (s + 1) * bps)];
        for (int64_t i = s * (int64_t)sps; i < (s + 1) * (int64_t)sps; ++i) raw[i] = v;
    }
    const int64_t off = (n >= m) ? (m - 1) / 2 : (n - 1) / 2;        /* "same": centred on the longer operand */
    for (int64_t i = 0; i < n; ++i) {
        if (freq_in) { f[i] = freq_in[i]; continue; }
        const int64_t k = i + off;
        const int64_t lo = k - m + 1 > 0 ? k - m + 1 : 0, hi = k < n - 1 ? k : n - 1;
        double acc = 0.0;
        for (int64_t j = lo; j <= hi; ++j) acc += (double)raw[j] * (double)gfir[k - j];
        f[i] = (float)acc;
    }
    ph[0] = carrier_phase;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const float t = arange_f32(start, i) / sample_rate;
        const float df = f[i] - f[i + 1];
        ph[i + 1] = (float)((((2.0 * M_PI) * (double)t) * (double)df) + (double)ph[i]);
    }
    for (int64_t i = 0; i < n; ++i) {
        const float t = ((float)(i + (int64_t)start)) / sample_rate;
        const float arg = (float)(((((2.0 * M_PI) * f[i]) * t) + ph[i]) + 0.0f);
        store_iq(out, dt, 2 * i, carrier_amplitude * cosf(arg));
        store_iq(out, dt, 2 * i + 1, carrier_amplitude * sinf(arg));
    }
    if (freq_out) memcpy(freq_out, f, sizeof(float) * (size_t)n);
    if (phase_out) memcpy(phase_out, ph, sizeof(float) * (size_t)n);
    free(raw); free(f); free(ph);
    return total_samples;
}
