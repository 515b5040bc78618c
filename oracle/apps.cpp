// oracle/apps.cpp -- CPU restatement of the application-side steps that bracket a lele model run.
// TEST INFRASTRUCTURE ONLY (see oracle.h): the product never links or calls this file.
//
// Follows, statement by statement:
//   /root/reference/examples/sensevoice/src/tokenizer.rs:37-86   (decode_greedy: arg-max, blank / special-token filter)
//   /root/reference/examples/yolo26n-seg/src/image.rs:62-111     (Image::preprocess + nearest resize)
//   /root/reference/examples/yolo26n-seg/src/image.rs:127-265    (postprocess_segmentation)
//   /root/reference/examples/silero/src/main.rs:151-228          (VAD segment collection and merging)
// Parity unpinned by golden vectors: the reference's tests hold no fixtures for these steps (they are example-binary code);
// the restatement is line-for-line small and uses the same libm (glibc expf) the Rust binary calls.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "oracle.h"

// tokenizer.rs:50-71.  skip[id] != 0 stands for `token_id == 0 || (token.starts_with("<|") && token.ends_with("|>"))`;
// ids >= vocab are dropped (tokenizer.rs:63).  Returns the number of kept ids of each row in counts, ids packed in out.
extern "C" void orc_decode_greedy_ids(const float* logits, int64_t batch, int64_t steps, int64_t vocab, const uint8_t* skip,
                                      int64_t skip_len, int32_t* out, int32_t* counts) {
    for (int64_t b = 0; b < batch; ++b) {
        int32_t n = 0;
        for (int64_t t = 0; t < steps; ++t) {
            const float* row = logits + (b * steps + t) * vocab;
            int64_t best = 0;  // Iterator::max_by returns the LAST maximum; unwrap_or(0) on an empty row
            for (int64_t j = 1; j < vocab; ++j)
                if (!(row[j] < row[best])) best = j;
            if (best < skip_len) {
                if (skip[best]) continue;
                out[b * steps + n++] = (int32_t)best;
            }
        }
        counts[b] = n;
        for (int64_t t = n; t < steps; ++t) out[b * steps + t] = -1;
    }
}

// image.rs:84-105 then 69-79
extern "C" void orc_image_preprocess(const uint8_t* rgb, int64_t height, int64_t width, int64_t target, float* out) {
    std::vector<uint8_t> resized((size_t)target * target * 3);
    for (int64_t y = 0; y < target; ++y)
        for (int64_t x = 0; x < target; ++x) {
            int64_t sx = (int64_t)floorf(((float)x + 0.5f) * (float)width / (float)target);
            int64_t sy = (int64_t)floorf(((float)y + 0.5f) * (float)height / (float)target);
            sx = std::min(sx, width - 1);
            sy = std::min(sy, height - 1);
            for (int c = 0; c < 3; ++c) resized[(y * target + x) * 3 + c] = rgb[(sy * width + sx) * 3 + c];
        }
    for (int c = 0; c < 3; ++c)
        for (int64_t h = 0; h < target; ++h)
            for (int64_t w = 0; w < target; ++w)
                out[c * target * target + h * target + w] = (float)resized[(h * target + w) * 3 + c] / 255.0f;
}

// image.rs:127-265.  dets: [300][38] records (x1, y1, x2, y2, score, class id, 32 coefficients), returns the count.
extern "C" int32_t orc_yolo_seg_postprocess(const float* logits, const float* mask_features, int64_t mask_total, int64_t img_width,
                                            int64_t img_height, float threshold, int64_t num_classes, float* dets,
                                            uint8_t* mask_img) {
    const int NUM_QUERIES = 300, MASK_DIM = 32, LOGIT_LEN = 38;
    const int64_t mask_hw = mask_total / MASK_DIM;
    const int64_t mask_h = (int64_t)sqrtf((float)mask_hw), mask_w = mask_h;
    float scale_x = (float)img_width / 640.0f, scale_y = (float)img_height / 640.0f;
    int32_t n = 0;
    for (int i = 0; i < NUM_QUERIES; ++i) {
        const float* q = logits + i * LOGIT_LEN;
        const float score = q[4];
        if (score < threshold) continue;
        const float cf = q[5];
        int64_t class_id = cf > 0.0f ? (cf >= 9.2e18f ? INT64_MAX : (int64_t)cf) : 0;  // `as usize` saturates
        class_id = std::min(class_id, num_classes - 1);
        const float x1r = q[0], y1r = q[1], x2r = q[2], y2r = q[3];
        if (x2r <= x1r || y2r <= y1r) continue;
        float* d = dets + n * LOGIT_LEN;
        d[0] = fmaxf(x1r * scale_x, 0.0f);
        d[1] = fmaxf(y1r * scale_y, 0.0f);
        d[2] = fminf(x2r * scale_x, (float)img_width);
        d[3] = fminf(y2r * scale_y, (float)img_height);
        d[4] = score;
        d[5] = (float)class_id;
        for (int j = 0; j < MASK_DIM; ++j) d[6 + j] = q[6 + j];
        ++n;
    }
    memset(mask_img, 0, (size_t)(img_width * img_height));
    if (n == 0) return 0;
    scale_x = (float)mask_w / (float)img_width;
    scale_y = (float)mask_h / (float)img_height;
    std::vector<float> det_mask((size_t)(mask_h * mask_w));
    for (int32_t k = 0; k < n; ++k) {
        const float* d = dets + k * LOGIT_LEN;
        for (int64_t y = 0; y < mask_h; ++y)
            for (int64_t x = 0; x < mask_w; ++x) {
                float sum = 0.0f;
                for (int c = 0; c < MASK_DIM; ++c) sum += d[6 + c] * mask_features[c * mask_h * mask_w + y * mask_w + x];
                det_mask[y * mask_w + x] = 1.0f / (1.0f + expf(-sum));
            }
        for (int64_t iy = 0; iy < img_height; ++iy)
            for (int64_t ix = 0; ix < img_width; ++ix) {
                int64_t mx = (int64_t)floorf(((float)ix + 0.5f) * scale_x), my = (int64_t)floorf(((float)iy + 0.5f) * scale_y);
                mx = std::min(mx, mask_w - 1);
                my = std::min(my, mask_h - 1);
                const float mv = det_mask[my * mask_w + mx];
                const bool in_bbox = (float)ix >= d[0] && (float)ix <= d[2] && (float)iy >= d[1] && (float)iy <= d[3];
                if (in_bbox && mv > 0.5f)
                    if (mv * d[4] > 0.5f) mask_img[iy * img_width + ix] = 255;
            }
    }
    return n;
}

// silero main.rs:151-228.  probs: one speech probability per chunk.  segments: [max_segments][2] (start, end) in samples;
// returns the number of merged segments.
extern "C" int64_t orc_vad_segments(const float* probs, int64_t num_probs, int64_t chunk_size, int64_t padded_len, int64_t audio_len,
                                    uint32_t sample_rate, float threshold, float min_silence_ms, float min_speech_ms,
                                    float speech_pad_ms, float merge_gap_ms, int64_t* segments, int64_t max_segments) {
    auto ms_to_samples = [](float ms, uint32_t sr) { return (int64_t)roundf((float)sr * (ms / 1000.0f)); };
    const int64_t min_silence = std::max<int64_t>(ms_to_samples(min_silence_ms, sample_rate), 1);
    const int64_t min_speech = std::max<int64_t>(ms_to_samples(min_speech_ms, sample_rate), 1);
    const int64_t speech_pad = ms_to_samples(speech_pad_ms, sample_rate);
    const int64_t merge_gap = ms_to_samples(merge_gap_ms, sample_rate);
    std::vector<std::pair<int64_t, int64_t>> segs;
    bool triggered = false;
    int64_t curr_start = 0, silence_acc = 0;
    for (int64_t i = 0; i < num_probs; ++i) {
        const int64_t offset = i * chunk_size, frame_end = std::min(offset + chunk_size, padded_len);
        if (probs[i] >= threshold) {
            if (!triggered) {
                triggered = true;
                curr_start = offset > speech_pad ? offset - speech_pad : 0;  // saturating_sub
            }
            silence_acc = 0;
        } else if (triggered) {
            silence_acc += frame_end - offset;
            if (silence_acc >= min_silence) {
                int64_t end = std::min(frame_end + speech_pad, audio_len);
                if (end > curr_start && end - curr_start >= min_speech) segs.push_back({curr_start, end});
                triggered = false;
                silence_acc = 0;
            }
        }
    }
    if (triggered) {
        const int64_t end = audio_len;
        if (end > curr_start && end - curr_start >= min_speech) segs.push_back({curr_start, end});
    }
    std::stable_sort(segs.begin(), segs.end(), [](const auto& a, const auto& b) { return a.first < b.first; });  // sort_by_key is stable
    std::vector<std::pair<int64_t, int64_t>> merged;
    for (const auto& seg : segs) {
        if (!merged.empty()) {
            auto& last = merged.back();
            if (seg.first <= last.second) {
                if (seg.second > last.second) last.second = seg.second;
                continue;
            }
            const int64_t gap = seg.first > last.second ? seg.first - last.second : 0;
            if (gap <= merge_gap) {
                if (seg.second > last.second) last.second = seg.second;
                continue;
            }
        }
        merged.push_back(seg);
    }
    const int64_t n = std::min<int64_t>((int64_t)merged.size(), max_segments);
    for (int64_t i = 0; i < n; ++i) {
        segments[2 * i] = merged[i].first;
        segments[2 * i + 1] = merged[i].second;
    }
    return (int64_t)merged.size();
}
