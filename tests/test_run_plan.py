"""Host logic of the synthesis in runs (hpr.hip plan_wide_run through zen_hip_run_plan): which run length the large-hop pass
of HPRIOffline takes for a batch, and where it declines.  No device needed."""
import time

import pytest


@pytest.fixture(scope="module")
def z():
    import zen_amd
    return zen_amd


def test_config4_batch_fills_the_device_with_four_runs_per_clip(z):
    # BASELINE config 4 per GPU: 64 clips x 30 s at hop 4096 = 324 frames each, groups {P + R, H} at nfft 16384 (256 workgroup
    # slots): 256 workgroups of 82 frames x 2 outputs, then 256 of 82 x 1 -- every CU gets one of each
    run, busy = z.run_plan(324, 64, 16384, (2, 1))
    assert run == 81 and busy > 0.98
    # one output group alone (the harmonic output left out): equal workgroups, again a multiple of the slots
    run1, busy1 = z.run_plan(324, 64, 16384, (2,))
    assert run1 in (81, 41, 27) and busy1 > 0.95


def test_a_single_clip_keeps_the_per_frame_launches(z):
    for frames in (40, 324, 2048):
        run, busy = z.run_plan(frames, 1, 16384, (2, 1))
        assert run == 0 and busy < 0.92, (frames, run, busy)


def test_runs_are_at_least_eight_frames_and_cover_the_pass(z):
    for frames, streams, nfft in ((100, 200, 8192), (1000, 37, 16384), (64, 512, 4096), (9, 4096, 2048), (5000, 3, 16384)):
        run, busy = z.run_plan(frames, streams, nfft, (2, 1))
        assert 0.0 <= busy <= 1.0 + 1e-9
        assert run == 0 or 8 <= run <= frames, (frames, streams, nfft, run)


def test_huge_batches_are_planned_without_a_long_simulation(z):
    t0 = time.perf_counter()
    run, busy = z.run_plan(324, 100000, 16384, (2, 1))
    run2, busy2 = z.run_plan(60000, 2000, 16384, (2, 1))
    assert time.perf_counter() - t0 < 2.0
    assert busy == 1.0 and 8 <= run <= 324          # more streams than 16 x the slots: any split fills the device
    assert busy2 > 0.9 and 8 <= run2 <= 60000       # 4 000 equal workgroups per run count: a few long runs per stream


def test_bad_arguments_are_refused(z):
    for args in ((324, 64, 1024, (2, 1)), (324, 64, 16384, (3,)), (324, 64, 12345, (1,))):
        with pytest.raises(z.ZenHipError):
            z.run_plan(*args)
