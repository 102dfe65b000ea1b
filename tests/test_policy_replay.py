"""SURVEY.md 8f rows 2 + 3: corpus -> MonoTextData batches -> the aggressive inner loop + joint steps + the outer-loop policy,
replayed against a recorded run of the reference's own main() (tests/golden/make_golden_policy*.py)."""
import pytest

import parity_common as pc


def test_text_policy_replay_first_epochs_emulated(emu_backend, tmp_path):
    """The aggressive phase (13 inner loops with their windowed exits, the MI stop check) and the first plain epoch on the
    CI emulator; the whole 19-epoch run (learning-rate decay, best-checkpoint bookkeeping) is the GPU test below."""
    r = pc.check_policy_replay_text("cpu", tmp_path, max_epochs=2)
    assert r["inner_steps"] == 615


@pytest.mark.gpu
def test_text_policy_replay(hip_device, tmp_path):
    r = pc.check_policy_replay_text(hip_device, tmp_path)
    assert r["iterations"] == 247 and r["inner_steps"] == 615


@pytest.mark.gpu
def test_image_policy_replay(hip_device):
    """image.py:267-428 (Omniglot outer loop) against the recorded reference run; GPU only (one emulated PixelCNN step takes a
    minute on the CI box)."""
    r = pc.check_policy_replay_image(hip_device)
    assert r["inner_steps"] > 0


def test_text_policy_replay_free_running_first_epoch_emulated(emu_backend, tmp_path):
    """The first epoch of the free-running replay (decoder dropout 0.5 fed from the recorded keep-masks, no re-synchronisation)
    on the CI emulator; all three epochs are the GPU test below."""
    r = pc.check_policy_replay_text("cpu", tmp_path, max_epochs=1, free_running=True)
    assert r["iterations"] == 13


@pytest.mark.gpu
def test_text_policy_replay_free_running(hip_device, tmp_path):
    """Three epochs of the reference's text.main() with dec_dropout 0.5, never re-synchronised: every one of the 1080 inner encoder
    steps runs on the weights our own previous steps left, and the windowed exits, the batch picks, the end of the aggressive
    phase and the best-checkpoint updates still equal the recorded run's."""
    r = pc.check_policy_replay_text(hip_device, tmp_path, free_running=True)
    assert r["iterations"] == 39 and r["inner_steps"] == 1080
