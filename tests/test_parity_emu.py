"""Host logic + kernel index logic on the CPU stand-in for the HIP runtime (tests/emu) against the checker.
This is NOT the product path (that is test_parity_gpu.py on a real gfx950); it runs in the GPU-less container."""
import pytest

import parity_cases as pc
import scenarios


@pytest.mark.parametrize("name", ["identity_mono_44k", "stretch_1p5_stereo", "pitch_p12_stereo", "cheaper_96k_3ch",
                                  "flush_short_stereo", "cheaper_48k_stereo", "eight_channels_1p5", "default_192k_mono", "cheaper_192k_mono"])
def test_golden(emu, ref, name):
    pc.case_golden(emu, ref, name)


def test_api_surface(emu, ref):
    pc.case_api_surface(emu, ref)


def test_split_mode(emu, ref):
    pc.case_split_mode(emu, ref)


def test_pitch_and_formants(emu, ref):
    pc.case_pitch_and_formants(emu, ref)


def test_large_plan_mapped(emu, ref):
    pc.case_large_plan_mapped(emu, ref)


def test_silence(emu, ref):
    pc.case_silence(emu, ref)


def test_channels(emu, ref):
    pc.case_channels(emu, ref)
    pc.case_channels(emu, ref, channel_counts=(9, 12))  # beyond the fused kernels' eight: kPredictB + kChain (16 on the device too)


def test_batch_ragged(emu, ref):
    pc.case_batch_ragged(emu, ref)


def test_random_time_factor(emu, ref):
    pc.case_random_time_factor(emu, ref)


def test_sub_batches(emu, ref, monkeypatch):
    pc.case_sub_batches(emu, ref, monkeypatch)


def test_cmd_main_flow_config1(emu, ref):
    pc.case_cmd_main_flow(emu, ref)


def test_realtime_quanta(emu, ref):
    pc.case_realtime_quanta(emu, ref)


def test_teacher_forced_small(emu, ref):
    """Teacher-forced single hops (SURVEY App. D.2 i) at the small geometry: plain, pitch-mapped, formants, 3 channels."""
    print(pc.case_teacher_forced(emu, ref, pc.SMALL, 2, 1.5, "forced plain"))
    print(pc.case_teacher_forced(emu, ref, pc.SMALL, 2, 1.0, "forced pitch", setup=lambda o: o.setTransposeSemitones(12, 8000/48000)))
    print(pc.case_teacher_forced(emu, ref, pc.SMALL, 2, 0.75, "forced formants",
                                 setup=lambda o: (o.setTransposeSemitones(4, 8000/48000), o.setFormantFactor(1, True), o.setFormantBase(200/48000))))
    print(pc.case_teacher_forced(emu, ref, pc.SMALL_SPLIT, 3, 1.2, "forced split 3ch", setup=lambda o: o.setTransposeSemitones(-5, 0)))


def test_hop_magnitudes_small(emu, ref):
    print(pc.case_hop_magnitudes(emu, ref, pc.SMALL, 2, 1.5, "magnitudes plain", hops=24))
    print(pc.case_hop_magnitudes(emu, ref, pc.SMALL, 2, 1.0, "magnitudes pitch", hops=24, setup=lambda o: o.setTransposeSemitones(12, 8000/48000)))


def test_gather_pass_shapes(emu, monkeypatch):
    pc.case_gather_pass_shapes(emu, monkeypatch, n=6000)


def test_vocn_writer_forms(emu, monkeypatch):
    pc.case_vocn_writer_forms(emu, monkeypatch, channel_counts=(3, 8), n=6000)


def test_fused_equals_unfused(emu, monkeypatch):
    pc.case_fused_equals_unfused(emu, monkeypatch, channel_counts=(2, 3, 5))


def test_feed_fusion_equals_separate(emu, monkeypatch):
    pc.case_feed_fusion_equals_separate(emu, monkeypatch)
    pc.case_feed_fusion_equals_separate(emu, monkeypatch, channel_counts=(2,), formants=True)
    pc.case_feed_fusion_equals_separate(emu, monkeypatch, channel_counts=(2, 3), formants=True, bases_given=True)  # the one-pass form (round 6)


def test_single_hop_chunks(emu):
    pc.case_single_hop_chunks(emu)
    pc.case_single_hop_chunks(emu, channel_counts=(2,), setup=lambda b: b.setTransposeSemitones(5, 0.2))


def test_half_state(emu, ref):
    print(pc.case_half_state(emu, ref, pc.SMALL, 2, 1.5, "half plain"))
    print(pc.case_half_state(emu, ref, pc.SMALL_SPLIT, 3, 1.2, "half 3ch split pitch", setup=lambda o: o.setTransposeSemitones(-5, 0)))


def test_freq_map_tables_are_per_stream(emu):
    """setFreqMap in table form: every stream keeps its own table, whatever its length (and whatever the other streams' lengths).
    Two tables of different length that describe the SAME linear map give the same
    output, and neither disturbs a third stream that has no map (the reference's instances share nothing)."""
    import numpy as np
    from conftest import package, synth_input
    pkg = package()
    C, n = 2, 6000
    x = synth_input(0, C, n, 48000) + 0.4*synth_input(4, C, n, 48000)
    xs = np.stack([x, x, x])
    t64 = np.array([(i + 0.5)/128*1.3 for i in range(64)], np.float32)
    t160 = np.array([(i + 0.5)/320*1.3 for i in range(160)], np.float32)
    b = pkg.StretchBatch(3, C, block=512, interval=128, lib=emu)
    b.setFreqMapTable(t64, stream=0)
    b.setFreqMapTable(t160, stream=1)   # another length: must not clear stream 0's map
    y = b.process(xs, n)
    b.close()
    plain = pkg.StretchBatch(1, C, block=512, interval=128, lib=emu)
    y_plain = plain.process(x[None], n)
    plain.close()
    one = pkg.StretchBatch(1, C, block=512, interval=128, lib=emu)
    one.setFreqMapTable(t64)
    y_one = one.process(x[None], n)
    one.close()
    # stream 0 kept its map, knot for knot: the longer table given to stream 1 afterwards does not touch it (include/smst.h)
    assert np.array_equal(np.asarray(y[0]), np.asarray(y_one[0]))
    assert np.array_equal(y[2], y_plain[0])                     # stream 2 has none
    err = np.sqrt(np.mean((y[1] - y[0])**2)/np.mean(y[0]**2))   # same linear map, resampled: same result up to the table's own rounding
    assert err < 1e-3, err
    assert np.sqrt(np.mean((y[0] - y_plain[0])**2)/np.mean(y_plain[0]**2)) > 0.1  # and the map does something


def test_packed_complex_helpers_emu(emu):
    """tests/emu/smst_complex.h (the CPU stand-in's twin of the product's inline-assembly header) follows the same formulas."""
    pc.case_complex_helpers(emu)


def test_fast_fft_close_to_generic_emu(emu, monkeypatch):
    print(pc.case_fast_fft_close_to_generic(emu, monkeypatch, presets=(("cheaper", 48000), ("default", 96000)), seconds=0.45))


def test_random_time_factor_seeds_emu(emu, ref):
    print(pc.case_random_time_factor_seeds(emu, ref, streams=3, seconds=0.4, level_tol=0.05))


def test_clone_emu(emu):
    pc.case_clone(emu)


def test_map_table_lengths_emu(emu):
    pc.case_map_table_lengths(emu)


def test_debug_map_is_of_the_last_call_emu(emu):
    pc.case_debug_map_is_of_the_last_call(emu)


def test_split_mid_interval_flush_emu(emu, ref):
    print(pc.case_split_mid_interval_flush(emu, ref))


@pytest.mark.parametrize("variant", ["L6", "L7", "L8", "L6_3ch", "no_single_hop", "no_single_hop_3ch"])
def test_split_mid_interval_flush_wide_emu(emu, ref, monkeypatch, variant):
    """HopDesc.startBin in every recurrence form, not only kVocoderOne"""
    if variant.startswith("no_single_hop"):
        monkeypatch.setenv("SMST_NO_SINGLE_HOP", "1")
        print(pc.case_split_mid_interval_flush_wide(emu, ref, channels=3 if variant.endswith("3ch") else 2, block=512))
        return
    block = {"L6": 768, "L7": 896, "L8": 1024}[variant[:2]]
    print(pc.case_split_mid_interval_flush_wide(emu, ref, channels=3 if variant.endswith("3ch") else 2, block=block))


def test_across_equals_single_hop_emu(emu, monkeypatch):
    pc.case_across_equals_single_hop(emu, monkeypatch, streams=11)
    pc.case_across_equals_single_hop(emu, monkeypatch, streams=5, channel_counts=(2,), setup=lambda b: b.setTransposeSemitones(5, 0.2))


def test_carried_emit_equals_copy_emu(emu, monkeypatch):
    pc.case_carried_emit_equals_copy(emu, monkeypatch)


def test_random_call_sequences_emu(emu, ref):
    pc.case_random_call_sequences(emu, ref, seeds=range(6))


def test_random_call_sequences_split_emu(emu, ref):
    """the same walks in split-computation mode: parameter changes, flushes, seeks and resets fall between interval boundaries"""
    print(pc.case_random_call_sequences(emu, ref, seeds=range(100, 108), cfg=pc.SMALL_SPLIT))


@pytest.mark.parametrize("geometry", scenarios.split_event_geometries())
def test_split_events_golden_emu(emu, ref, geometry):
    print(pc.case_split_events_golden(emu, ref, geometry))


def test_split_events_vs_checker_emu(emu, ref):
    print(pc.case_split_events_vs_checker(emu, ref))


def test_random_time_factor_parity_emu(emu, ref):
    print(pc.case_random_time_factor_parity(emu, ref))


def test_fft_teams_equals_per_frame_emu(emu, monkeypatch):
    pc.case_fft_teams_equals_per_frame(emu, monkeypatch, presets=(("cheaper", 48000),), seconds=0.45, streams=3)
    # 44.1 kHz: the window's halves do not end on element-slot boundaries (kAnalyseTeams<..., SLOTS = false>)
    pc.case_fft_teams_equals_per_frame(emu, monkeypatch, presets=(("default", 44100),), seconds=0.4, streams=2)


def test_synth_emit_equals_two_kernels_emu(emu, monkeypatch):
    pc.case_synth_emit_equals_two_kernels(emu, monkeypatch, presets=(("cheaper", 48000),), seconds=0.45, streams=3)
    pc.case_synth_emit_equals_two_kernels(emu, monkeypatch, presets=(("default", 44100),), seconds=0.4, streams=2, channels=1, splits=(True,))


def test_hop_magnitudes_per_stream_parameters_small(emu, ref):
    """The per-stream form of the phase-free instrument (every stream its own stretch factor and transposition, ragged input spans in
    one batched call) at the small geometry, 3 channels, split mode -- the GPU suite runs it on BASELINE config 5 as named."""
    print(pc.case_hop_magnitudes(emu, ref, pc.SMALL_SPLIT, 3, [0.8, 1.0, 1.37], "magnitudes per stream", hops=30, streams=(0, 1, 2), tol=2e-4,
                                 semitones=[-7.0, 3.5, 11.0]))


def test_reconfigure_keeps_random_engine(emu, ref):
    pc.case_reconfigure_keeps_random_engine(emu, ref)


def test_split_batch_events_emu(emu, monkeypatch):
    """split computation in a batch: streams at different offsets of their intervals, per-stream setters and flushes == every stream alone"""
    pc.case_split_batch_events(emu, monkeypatch)


def test_split_dropped_block_random_engine_emu(emu, ref):
    print(pc.case_split_dropped_block_random_engine(emu, ref))


def test_api_surface_and_realtime_quanta_split_emu(emu, ref):
    """seek / ragged chunks / flush / outputSeek / exact and the AudioWorklet calling patterns in split-computation mode"""
    pc.case_api_surface(emu, ref, cfg=pc.SMALL_SPLIT)
    pc.case_realtime_quanta(emu, ref, cfg=pc.SMALL_SPLIT)


def test_split_freq_map_mid_interval_emu(emu, ref):
    print(pc.case_split_freq_map_mid_interval(emu, ref))


def test_continuous_equals_tiled_emu(emu, monkeypatch):
    print(pc.case_continuous_equals_tiled(emu, monkeypatch))


def test_formant_stages_emu(emu, ref, monkeypatch):
    print(pc.case_formant_stages(emu, ref, monkeypatch, pc.SMALL, hops=14))
