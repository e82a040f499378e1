"""A second, independent restatement of the reference's ENCODER in straight-line pure Python (test infrastructure).

Written from rafaelcaricio/qoaudio src/lib.rs:333-398 (encoder object), :441-493 (frame), :495-596 (slice search in the
order the crate executes it: first-sample pass, stable insertion sort, both early exits), :602-617 (frame size, qoa_div),
:797-864 (LMS, tables) -- deliberately NOT from oracle/qoa_oracle.c and sharing no code with it: Python integers with
explicit 32-bit wrapping instead of C integer types, the fused quant+dequant table built the way :831-845 builds it.
tests/test_oracle_pyref.py runs it against the C oracle (both search forms) on penalty-active, tie-heavy and wrap-heavy
inputs: two restatements by different routes agreeing is what backs the oracle where the reference's fixture does not
reach (oracle/README.md)."""

SLICE_LEN, LMS_LEN, FRAME_LEN, MAGIC = 20, 4, 256 * 20, 0x716F6166

QUANT_TAB = [7, 7, 7, 5, 5, 3, 3, 1, 0, 0, 2, 2, 4, 4, 6, 6, 6]                      # :22
RECIPROCAL_TAB = [65536, 9363, 3121, 1457, 781, 475, 311, 216, 156, 117, 90, 71, 57, 47, 39, 32]  # :25-27
DEQUANT_TAB = [                                                                       # :847-864
    [1, -1, 3, -3, 5, -5, 7, -7], [5, -5, 18, -18, 32, -32, 49, -49], [16, -16, 53, -53, 95, -95, 147, -147],
    [34, -34, 113, -113, 203, -203, 315, -315], [63, -63, 210, -210, 378, -378, 588, -588],
    [104, -104, 345, -345, 621, -621, 966, -966], [158, -158, 528, -528, 950, -950, 1477, -1477],
    [228, -228, 760, -760, 1368, -1368, 2128, -2128], [316, -316, 1053, -1053, 1895, -1895, 2947, -2947],
    [422, -422, 1405, -1405, 2529, -2529, 3934, -3934], [548, -548, 1828, -1828, 3290, -3290, 5117, -5117],
    [696, -696, 2320, -2320, 4176, -4176, 6496, -6496], [868, -868, 2893, -2893, 5207, -5207, 8099, -8099],
    [1064, -1064, 3548, -3548, 6386, -6386, 9933, -9933], [1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005],
    [1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336]]
# :831-845: (quantized << 32) | (dequantized as u32)
QUANT_DEQUANT_TAB = [[(QUANT_TAB[i] << 32) | (DEQUANT_TAB[sf][QUANT_TAB[i]] & 0xFFFFFFFF) for i in range(17)] for sf in range(16)]
U64_MAX = (1 << 64) - 1


def i32(v):
    """two's-complement wrap to i32 (Rust wrapping_* / `as i32`)"""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def qoa_div(v, sf):  # :613-617
    n = i32(i32(v * RECIPROCAL_TAB[sf]) + (1 << 15)) >> 16
    return n + ((v > 0) - (v < 0)) - ((n > 0) - (n < 0))


class Lms:
    __slots__ = ("history", "weights")

    def __init__(self, history, weights):
        self.history, self.weights = list(history), list(weights)

    def copy(self):
        return Lms(self.history, self.weights)

    def predict_and_penalty_sq(self):  # :808-818
        w, h = self.weights, self.history
        p01 = i32(i32(w[0] * h[0]) + i32(w[1] * h[1]))
        p23 = i32(i32(w[2] * h[2]) + i32(w[3] * h[3]))
        prediction = i32(p01 + p23) >> 13
        s01 = i32(i32(w[0] * w[0]) + i32(w[1] * w[1]))
        s23 = i32(i32(w[2] * w[2]) + i32(w[3] * w[3]))
        penalty = max((i32(s01 + s23) >> 18) - 0x8FF, 0)
        return prediction, penalty * penalty

    def update(self, sample_i16, residual):  # :821-828 (release build: `+=` wraps)
        delta = residual >> 4
        for i in range(4):
            self.weights[i] = i32(self.weights[i] + (-delta if self.history[i] < 0 else delta))
        self.history = [self.history[1], self.history[2], self.history[3], sample_i16]


def clamp(v, lo, hi):
    return lo if v < lo else hi if v > hi else v


def as_i16(v):
    v &= 0xFFFF
    return v - 0x10000 if v & 0x8000 else v


def encode_slice(channel_lms, sample_data, slice_start, slice_end, channels):  # :495-596
    samples = [sample_data[si] for si in range(slice_start, slice_end, channels)][:SLICE_LEN]
    slice_len = len(samples)
    best_rank, best_slice, best_scalefactor, best_lms = U64_MAX, 0, 0, Lms([0] * 4, [0] * 4)
    first_predicted, first_penalty_sq = channel_lms.predict_and_penalty_sq()
    first_sample = samples[0]
    first_residual = first_sample - first_predicted
    first_results = [None] * 16
    sf_order = []
    for sf in range(16):
        scaled = qoa_div(first_residual, sf)
        packed = QUANT_DEQUANT_TAB[sf][clamp(scaled, -8, 8) + 8]
        quantized, dequantized = packed >> 32, i32(packed)
        reconstructed = clamp(first_predicted + dequantized, -32768, 32767)
        error = first_sample - reconstructed
        rank = error * error + first_penalty_sq
        first_results[sf] = (packed, reconstructed, quantized, rank)
        pos = len(sf_order)
        sf_order.append(sf)
        while pos > 0 and first_results[sf_order[pos - 1]][3] > rank:  # stable insertion sort (:536-541)
            sf_order[pos] = sf_order[pos - 1]
            pos -= 1
        sf_order[pos] = sf
    for scalefactor in sf_order:
        tab = QUANT_DEQUANT_TAB[scalefactor]
        lms = channel_lms.copy()
        slice_bits = scalefactor
        packed, reconstructed, quantized, first_rank = first_results[scalefactor]
        current_rank = first_rank
        if current_rank > best_rank:  # :555-557
            break
        lms.update(as_i16(reconstructed), i32(packed))
        slice_bits = (slice_bits << 3) | quantized
        valid = True
        for i in range(1, slice_len):
            sample = samples[i]
            predicted, penalty_sq = lms.predict_and_penalty_sq()
            residual = sample - predicted
            packed = tab[clamp(qoa_div(residual, scalefactor), -8, 8) + 8]
            quantized, dequantized = packed >> 32, i32(packed)
            reconstructed = clamp(predicted + dequantized, -32768, 32767)
            error = sample - reconstructed
            current_rank += error * error + penalty_sq
            if current_rank > best_rank:  # :578-581
                valid = False
                break
            lms.update(as_i16(reconstructed), dequantized)
            slice_bits = (slice_bits << 3) | quantized
        if valid and current_rank < best_rank:  # :587-592
            best_rank, best_slice, best_scalefactor, best_lms = current_rank, slice_bits, scalefactor, lms
    return best_slice & U64_MAX, best_scalefactor, best_lms


class Encoder:  # :85-91, :335-439
    def __init__(self, channels, sample_rate, samples):
        assert 1 <= channels <= 8 and sample_rate > 0 and samples > 0
        self.channels, self.sample_rate, self.samples = channels, sample_rate, samples
        self.lms = [Lms([0] * 4, [0, 0, -(1 << 13), 1 << 14]) for _ in range(channels)]

    def encode_frame_to_buf(self, sample_data):  # :441-493
        ch = self.channels
        frame_len = len(sample_data) // ch
        slices = -(-frame_len // SLICE_LEN)
        frame_size = (8 + LMS_LEN * 4 * ch + 8 * slices * ch) & 0xFFFF
        out = bytearray()
        header = (ch << 56) | (self.sample_rate << 32) | (frame_len << 16) | frame_size
        out += (header & U64_MAX).to_bytes(8, "big")
        for c in range(ch):
            history = weights = 0
            for i in range(LMS_LEN):
                history = (history << 16) | (self.lms[c].history[i] & 0xFFFF)
                weights = (weights << 16) | (self.lms[c].weights[i] & 0xFFFF)
            out += history.to_bytes(8, "big") + weights.to_bytes(8, "big")
        for sample_index in range(0, frame_len, SLICE_LEN):
            for c in range(ch):
                slice_len = min(frame_len - sample_index, SLICE_LEN)
                best_slice, _, best_lms = encode_slice(self.lms[c], sample_data, sample_index * ch + c,
                                                       (sample_index + slice_len) * ch + c, ch)
                self.lms[c] = best_lms
                if slice_len < SLICE_LEN:
                    best_slice = (best_slice << ((SLICE_LEN - slice_len) * 3)) & U64_MAX
                out += best_slice.to_bytes(8, "big")
        return bytes(out)

    def encode(self, sample_data):  # :367-398
        assert len(sample_data) == self.samples * self.channels
        out = bytearray(MAGIC.to_bytes(4, "big") + self.samples.to_bytes(4, "big"))
        sample_index = 0
        while sample_index < self.samples:
            frame_len = min(self.samples - sample_index, FRAME_LEN)
            out += self.encode_frame_to_buf(sample_data[sample_index * self.channels:(sample_index + frame_len) * self.channels])
            sample_index += frame_len
        return bytes(out)


def encode_all(sample_data, channels, sample_rate, samples):  # :760-763
    return Encoder(channels, sample_rate, samples).encode([int(v) for v in sample_data])
