"""Host sequencing + kernel index math on the GPU-less CI: the SAME kernel sources compiled with g++ against the
thread-level emulator (tests/emu), driven through the drop-in modules, against the golden fixtures.
This validates the code that will run on the MI355X, it is not a product path (the package refuses CPU tensors
unless a test installs the emulator backend)."""
import pytest
import torch

import parity_common as pc


@pytest.mark.parametrize("name,optim", [("text_small_refinit", "torch"), ("text_small_wide", "torch"), ("text_edge_T2", "torch"),
                                        ("text_small_wide", "lvae"), ("text_edge_T2", "lvae")])
def test_inner_step_matches_reference_fixture_emulated(emu_backend, name, optim):
    pc.check_step_against_fixture(name, "cpu", optim=optim)


def test_dropin_backward_keeps_autograd_accumulation_emulated(emu_backend):
    pc.check_grad_accumulation_semantics("cpu")


def test_dropin_backward_under_autograd_grad_and_frozen_modules_emulated(emu_backend):
    pc.check_autograd_grad_and_frozen_modules("cpu")


def test_cpu_tensors_refused_without_test_backend():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import install as emu_install
    from vae_lagging_encoder_amd import _lib, engine
    saved = emu_install.install(None)
    try:
        with pytest.raises(_lib.LvaeError):
            engine.backend_for(torch.device("cpu"))
    finally:
        emu_install.install(saved)


def test_fused_trainer_trajectory_emulated(emu_backend):
    pc.check_trajectory_against_fixture("cpu")


def test_inner_loop_exit_logic_emulated(emu_backend):
    steps = pc.check_inner_loop_exit_logic("cpu", window=2, max_iter=4)
    assert 1 <= steps < 4


def test_bf16_native_operands_equal_on_the_fly_emulated(emu_backend):
    pc.check_bf16_native_operands_equal_on_the_fly("cpu", V=133, ni=16, H=24, nz=4, B=5, T=6)


def test_update_both_and_fixed_k_emulated(emu_backend):
    pc.check_update_both_and_fixed_k("cpu")


def test_eval_statistics_against_reference_fixture_emulated(emu_backend):
    pc.check_eval_against_fixture("cpu")


def test_generation_against_reference_fixture_emulated(emu_backend):
    pc.check_generation_against_fixture("cpu")


def test_image_step_fused_emulated(emu_backend):
    """The Omniglot inner step (ResNet encoder + PixelCNN decoder, direct 32 -> 32 convolutions, BN, Adam) against the reference
    fixture on the emulator build of the same kernel sources (~1 min)."""
    pc.check_image_step_fused("image_b6", "cpu")


def test_weight_images_follow_rebound_parameters_emulated(emu_backend):
    pc.check_weight_images_follow_rebound_parameters("cpu")


def test_token_sort_cache_follows_the_batch_emulated(emu_backend):
    pc.check_token_sort_cache_follows_the_batch("cpu")


def test_pixelcnn_incremental_sampling_emulated(emu_backend):
    """Pixel-at-a-time sampling == full forward, bit for bit, on the emulator build (one image; the 784-full-pass comparison is the
    GPU test)."""
    pc.check_pixelcnn_incremental_sampling("cpu", B=1, compare_full_path=False)


@pytest.mark.parametrize("fault_at,rungs_down", [((2,), 1), ((0,), 2), ((1, 3), 2)])
def test_voided_steps_are_replayed_down_the_ladder_emulated(emu_backend, fault_at, rungs_down):
    """The transaction gate of the fused step (lv_clip_norm2_txn_f32 + lv_sgd_step_txn_f32) and the trainer's replay, with the
    status words a timed-out persistent launch would leave set by hand: the weights equal a run that never saw the fault."""
    pc.check_transactional_recovery("cpu", fault_at=fault_at, rungs_down=rungs_down)


@pytest.mark.parametrize("name,m", [("text_small_wide", 2), ("text_mid", 4), ("text_small_wide", 4)])
def test_micro_batches_give_the_whole_batch_gradient_emulated(emu_backend, name, m):
    pc.check_micro_batches_against_fixture(name, "cpu", m)


@pytest.mark.parametrize("shape", [(97, 12, 20, 4, 6, 7), (300, 512, 16, 4, 8, 40)])
def test_norm_folding_is_the_same_norm_emulated(emu_backend, shape):
    """trainer._plan_fold on the CI emulator: the embedding tables' sums of squares from the scatter's own pass (the second shape
    has ni = 512 and token runs long enough for the 8-group scatter kernel)."""
    V, ni, H, nz, B, T = shape
    pc.check_fold_norm("cpu", V, ni, H, nz, B, T)
    pc.check_fold_norm("cpu", V, ni, H, nz, B, T, decoder_grads="norm")


def test_voided_steps_with_norm_only_decoder_gradients_emulated(emu_backend):
    """The transaction gate and decoder_grads="norm" together: a voided run of steps is replayed and lands, bit for bit, where a
    fault-free run with the same option does (the joint decoder step in the sequence needs -- and gets -- full decoder gradients)."""
    pc.check_transactional_recovery("cpu", fault_at=(1, 3), rungs_down=2, decoder_grads="norm")


def test_guarded_eval_repeats_a_pass_that_saw_a_timeout(emu_backend):
    """training.guarded_eval: an evaluation pass is a forward-only use of the engines -- no transaction gate -- so a
    hand-off timeout inside it is looked for afterwards; the engines go one rung down, the event is logged, the pass runs again."""
    from helpers import build_vae
    from vae_lagging_encoder_amd import engine as eng
    from vae_lagging_encoder_amd.training import guarded_eval
    vae = build_vae(53, 8, 16, 4, "cpu", seed=1)
    for m in (vae.encoder, vae.decoder):
        m._hip.ensure(torch.device("cpu"))
    calls, msgs = [], []

    def fn():
        calls.append(1)
        if len(calls) == 1:
            vae.decoder._hip.status.fill_(237)          # what a timed-out persistent BPTT launch would leave
        return len(calls)
    assert guarded_eval(vae, fn, log=msgs.append) == 2
    assert [eng.persist_rung(m._hip) for m in (vae.encoder, vae.decoder)] == [1, 1]
    assert int(vae.decoder._hip.status.item()) == 0 and len(msgs) == 1 and "ladder rung 1" in msgs[0]
    assert guarded_eval(vae, lambda: "fine", log=msgs.append) == "fine" and len(msgs) == 1
