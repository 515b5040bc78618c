"""Host-side halves of the application steps around a lele model run.

The device halves are in csrc/app.hip (`kernels.wav_to_f32`, `argmax_last`, `token_filter`, `image_preprocess`,
`yolo_seg_postprocess`); what remains on the host is string handling and a tiny sequential state machine:

  * `special_token_mask`  -- the per-vocabulary skip flags of examples/sensevoice/src/tokenizer.rs:63-69
  * `detokenize`          -- tokenizer.rs:74-81 (join, sentencepiece underscore -> space, trim)
  * `vad_segments`        -- examples/silero/src/main.rs:151-228 (speech segments from per-chunk probabilities)
"""
import numpy as np


def special_token_mask(id_to_token):
    """skip[id] = 1 for the blank (id 0) and every "<|...|>" token (tokenizer.rs:66)"""
    skip = np.zeros(len(id_to_token), np.uint8)
    for i, tok in enumerate(id_to_token):
        if i == 0 or (tok.startswith("<|") and tok.endswith("|>")):
            skip[i] = 1
    return skip


def detokenize(ids, id_to_token):
    """tokenizer.rs:74-81: kept ids (as produced by kernels.token_filter, -1 padding ignored) -> text"""
    text = "".join(id_to_token[int(i)] for i in ids if int(i) >= 0)
    return text.replace("▁", " ").strip()


def _ms_to_samples(ms, sample_rate):
    # ((sr as f32) * (ms / 1000.0)).round() as usize -- f32 arithmetic, round half away from zero
    v = np.float32(sample_rate) * (np.float32(ms) / np.float32(1000.0))
    return int(np.floor(np.abs(v) + np.float32(0.5)) * np.sign(v))


def vad_segments(probs, chunk_size, padded_len, audio_len, sample_rate=16000, threshold=0.3, min_silence_ms=200.0,
                 min_speech_ms=400.0, speech_pad_ms=120.0, merge_gap_ms=200.0):
    """examples/silero/src/main.rs:151-228 with VadConfig::default (main.rs:18-28) -> [(start, end)] in samples"""
    min_silence = max(_ms_to_samples(min_silence_ms, sample_rate), 1)
    min_speech = max(_ms_to_samples(min_speech_ms, sample_rate), 1)
    speech_pad = _ms_to_samples(speech_pad_ms, sample_rate)
    merge_gap = _ms_to_samples(merge_gap_ms, sample_rate)
    thr = np.float32(threshold)
    segments, triggered, start, silence = [], False, 0, 0
    for i, prob in enumerate(np.asarray(probs, np.float32).reshape(-1)):
        offset = i * chunk_size
        frame_end = min(offset + chunk_size, padded_len)
        if prob >= thr:
            if not triggered:
                triggered, start = True, max(offset - speech_pad, 0)
            silence = 0
        elif triggered:
            silence += frame_end - offset
            if silence >= min_silence:
                end = min(frame_end + speech_pad, audio_len)
                if end > start and end - start >= min_speech:
                    segments.append([start, end])
                triggered, silence = False, 0
    if triggered and audio_len > start and audio_len - start >= min_speech:
        segments.append([start, audio_len])
    merged = []
    for seg in sorted(segments, key=lambda s: s[0]):  # stable, as sort_by_key
        if merged:
            last = merged[-1]
            if seg[0] <= last[1] or max(seg[0] - last[1], 0) <= merge_gap:
                last[1] = max(last[1], seg[1])
                continue
        merged.append(list(seg))
    return [(a, b) for a, b in merged]
