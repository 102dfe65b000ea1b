"""Outer training loop of the reference's text.py, as a reusable driver around the fused inner loop.

text.py:325-510, restated: per epoch a random permutation of the equal-length batches; per batch the KL weight anneals
linearly (kl_start -> 1 over `warm_up` epochs), the AGGRESSIVE inner loop runs while the flag is set (encoder-only steps until
the windowed loss stops improving: AggressiveTextTrainer.inner_loop = text.py:366-400), then one joint step (decoder only while
aggressive, encoder + decoder afterwards, text.py:407-424).  Once per epoch worth of iterations the mutual information on the
validation set decides whether aggressive training goes on ("STOP BURNING" when it drops, text.py:447-455).  Epoch end:
validation loss / NLL / KL / PPL / MI / active units, best-checkpoint bookkeeping, learning-rate decay (x0.5 after
`decay_epoch` = 2 epochs without improvement once epoch >= 15, reload the best weights, stop after `max_decay` = 5 decays,
text.py:463-489), a test-set pass every `test_nepoch` epochs.  The arithmetic is the HIP hot path throughout; the policy is
plain host code with the reference's constants and comparison operators.
"""
import copy
import math
import time

import numpy as np
import torch

from . import engine as _eng
from . import evaluation as E
from .trainer import AggressiveImageTrainer, AggressiveTextTrainer

CLIP_GRAD = 5.0          # text.py:17-20
DECAY_EPOCH = 2
LR_DECAY = 0.5
MAX_DECAY = 5


def guarded_eval(vae, fn, log=print):
    """Run an evaluation pass -- a forward-only use of the engines: there is no transaction gate in it -- and look at the engines'
    persistent-launch status words afterwards (one host read each; every evaluation ends in host reads anyway).  A hand-off
    timeout during the pass would have left garbage in its statistics, which drive best-state selection, learning-rate decay and
    the end of aggressive training: the engines are moved one rung down the fallback ladder (engine.demote_persistent), the event
    is logged, and the pass is run again.  Engines without persistent launches (the image model's) have no status word."""
    engines = [e for e in (getattr(vae.encoder, "_hip", None), getattr(vae.decoder, "_hip", None))
               if e is not None and getattr(e, "status", None) is not None]
    for _ in range(3):
        out = fn()
        bad = [e for e in engines if int(e.status.item()) != 0]
        if not bad:
            return out
        for e in engines:          # both engines take the same rung, as the trainers' _settle does
            if _eng.persist_rung(e) < 2:
                _eng.demote_persistent(e)
            else:
                _eng.reset_persistent_status(e)
        log("persistent LSTM launch timed out during an evaluation pass: engines moved to ladder rung %d (%s), pass repeated" % (
            max(_eng.persist_rung(e) for e in engines), _eng.PERSIST_RUNGS[max(_eng.persist_rung(e) for e in engines)]))
    raise _eng._lib.LvaeError("an evaluation pass kept reporting persistent-launch timeouts on every rung of the fallback ladder")


class TextTrainingLoop(object):
    """args fields read (text.py's argparse names): kl_start, warm_up, batch_size, epochs, aggressive, nsamples, test_nepoch,
    iw_nsamples, momentum (must be 0: the fused step is plain SGD, the reference default)."""

    def __init__(self, vae, train_batches, val_batches, test_batches, args, n_train_sentences=None, trainer=None,
                 log=print, np_rng=None, seed=783435, noise_fn=None, epoch_hook=None):
        """noise_fn(x) -> (eps, mask_in, mask_out): injects the random draws of every training step (parity replays of a
        recorded reference run, tests/test_policy_replay.py); None draws them on the device.
        epoch_hook(loop, epoch): called at the top of every epoch, before its batch permutation is drawn (checkpoint / resume
        policies; the replay test re-synchronises the weights there)."""
        self.vae, self.args, self.log = vae, args, log
        self.noise_fn = noise_fn
        self.epoch_hook = epoch_hook
        self.train_batches, self.val_batches, self.test_batches = train_batches, val_batches, test_batches
        if getattr(args, "momentum", 0) != 0:
            raise ValueError("the fused driver implements optim.SGD(momentum=0), the reference's default (text.py:325-326)")
        self.trainer = trainer if trainer is not None else AggressiveTextTrainer(vae, lr=1.0, clip=CLIP_GRAD, seed=seed)
        if hasattr(self.trainer, "prepare_batches"):
            self.trainer.prepare_batches(train_batches)          # per-batch index structures, built once with the batch list
        self.rng = np_rng if np_rng is not None else np.random
        n = n_train_sentences if n_train_sentences is not None else sum(int(b.shape[0]) for b in train_batches)
        self.n_train = n
        self.kl_weight = float(args.kl_start)
        self.anneal_rate = (1.0 - args.kl_start) / (args.warm_up * (n / args.batch_size))      # text.py:339
        self.opt = {"not_improved": 0, "lr": 1.0, "best_loss": 1e4}                              # text.py:251
        self.aggressive = bool(args.aggressive)
        self.iter_ = self.decay_cnt = 0
        self.pre_mi = 0.0
        self.best = {"loss": 1e4, "nll": 0.0, "kl": 0.0, "ppl": 0.0, "state": None}
        self.history = []           # one record per epoch
        self.iterations = []        # one record per outer iteration: what text.py's loop body did (batch, kl weight, inner steps)
        self.mi_checks = []         # (pre_mi, cur_mi) of every end-of-epoch aggressive check

    # -- pieces the tests drive directly --------------------------------------------------------------------------------------
    def _eval_mi_au(self):
        self.vae.eval()
        with torch.no_grad():
            mi, au = guarded_eval(self.vae, lambda: (E.calc_mi(self.vae, self.val_batches), E.calc_au(self.vae, self.val_batches)[0]), self.log)
        self.vae.train()
        return mi, au

    def check_aggressive(self):
        """text.py:447-455: called when a full epoch worth of iterations has passed while aggressive."""
        self.vae.eval()
        with torch.no_grad():
            cur_mi = guarded_eval(self.vae, lambda: E.calc_mi(self.vae, self.val_batches), self.log)
        self.vae.train()
        self.log("pre mi:%.4f. cur mi:%.4f" % (self.pre_mi, cur_mi))
        self.mi_checks.append((self.pre_mi, cur_mi))
        if cur_mi - self.pre_mi < 0:
            self.aggressive = False
            self.log("STOP BURNING")
        self.pre_mi = cur_mi

    def end_of_epoch(self, epoch, loss, nll, kl, ppl):
        """Best-checkpoint and learning-rate policy (text.py:463-489).  Returns True when training should stop."""
        if loss < self.best["loss"]:
            self.log("update best loss")
            self.best.update(loss=loss, nll=nll, kl=kl, ppl=ppl, state=copy.deepcopy(self.vae.state_dict()))
        if loss > self.opt["best_loss"]:
            self.opt["not_improved"] += 1
            if self.opt["not_improved"] >= DECAY_EPOCH and epoch >= 15:
                self.opt["best_loss"] = loss
                self.opt["not_improved"] = 0
                self.opt["lr"] *= LR_DECAY
                if self.best["state"] is not None:
                    self.vae.load_state_dict(self.best["state"])
                self.log("new lr: %f" % self.opt["lr"])
                self.decay_cnt += 1
                self.trainer.set_lr(self.opt["lr"])
        else:
            self.opt["not_improved"] = 0
            self.opt["best_loss"] = loss
        return self.decay_cnt == MAX_DECAY

    # -- the loop ---------------------------------------------------------------------------------------------------------------
    def run(self):
        args, tr = self.args, self.trainer
        log_niter = max(1, (self.n_train // args.batch_size) // 10)                               # text.py:263
        start = time.time()
        self.vae.train()
        for epoch in range(args.epochs):
            if self.epoch_hook is not None:
                self.epoch_hook(self, epoch)
            rep_rec = rep_kl = 0.0
            rep_sents = 0
            for i in self.rng.permutation(len(self.train_batches)):
                batch = self.train_batches[i]
                bsz, slen = batch.shape
                rep_sents += bsz
                self.kl_weight = min(1.0, self.kl_weight + self.anneal_rate)
                inner = 0
                if self.aggressive:
                    inner = tr.inner_loop(self.train_batches, batch, self.kl_weight, np_rng=self.rng, noise_fn=self.noise_fn)
                tr.reset_stats()
                tr.step(batch, self.kl_weight, noise=None if self.noise_fn is None else self.noise_fn(batch),
                        update="decoder" if self.aggressive else "both")
                st = tr.read_stats()
                rep_rec += st["rec_sum"]
                rep_kl += st["kl_sum"]
                self.iterations.append(dict(epoch=epoch, iter=self.iter_, batch=int(i), kl_weight=self.kl_weight,
                                            aggressive=self.aggressive, inner_steps=inner, rec_sum=st["rec_sum"], kl_sum=st["kl_sum"]))
                if self.iter_ % log_niter == 0:
                    train_loss = (rep_rec + rep_kl) / rep_sents
                    if self.aggressive or epoch == 0:
                        mi, au = self._eval_mi_au()
                        self.log("epoch: %d, iter: %d, avg_loss: %.4f, kl: %.4f, mi: %.4f, recon: %.4f,au %d, time elapsed %.2fs" % (
                            epoch, self.iter_, train_loss, rep_kl / rep_sents, mi, rep_rec / rep_sents, au, time.time() - start))
                    else:
                        self.log("epoch: %d, iter: %d, avg_loss: %.4f, kl: %.4f, recon: %.4f,time elapsed %.2fs" % (
                            epoch, self.iter_, train_loss, rep_kl / rep_sents, rep_rec / rep_sents, time.time() - start))
                    rep_rec = rep_kl = 0.0
                    rep_sents = 0
                self.iter_ += 1
                if self.aggressive and self.iter_ % len(self.train_batches) == 0:
                    self.check_aggressive()
            self.log("kl weight %.4f" % self.kl_weight)
            self.vae.eval()
            with torch.no_grad():
                # (np_rng: E.test draws from it; a repeated pass after a timeout draws again -- the recorded replays never time out)
                loss, nll, kl, ppl, mi = guarded_eval(
                    self.vae, lambda: E.test(self.vae, self.val_batches, "VAL", args, verbose=False, np_rng=self.rng), self.log)
                au = guarded_eval(self.vae, lambda: E.calc_au(self.vae, self.val_batches)[0], self.log)
            self.log("VAL --- avg_loss: %.4f, kl: %.4f, mi: %.4f, nll: %.4f, ppl: %.4f, %d active units" % (loss, kl, mi, nll, ppl, au))
            self.history.append(dict(epoch=epoch, loss=loss, nll=nll, kl=kl, ppl=ppl, mi=mi, au=au, aggressive=self.aggressive,
                                     kl_weight=self.kl_weight, lr=self.opt["lr"]))
            improved = loss < self.best["loss"]
            stop = self.end_of_epoch(epoch, loss, nll, kl, ppl)
            self.history[-1].update(best_updated=improved, lr_after=self.opt["lr"], decay_cnt=self.decay_cnt)
            if stop:
                break
            if epoch % getattr(args, "test_nepoch", 5) == 0 and self.test_batches:
                with torch.no_grad():
                    guarded_eval(self.vae, lambda: E.test(self.vae, self.test_batches, "TEST", args, verbose=False, np_rng=self.rng), self.log)
            self.vae.train()
        if self.best["state"] is not None:
            self.vae.load_state_dict(self.best["state"])
        return dict(best_loss=self.best["loss"], best_nll=self.best["nll"], best_kl=self.best["kl"], best_ppl=self.best["ppl"],
                    epochs=len(self.history), history=self.history)


class ImageTrainingLoop(object):
    """Outer training loop of the reference's image.py (image.py:267-428) around AggressiveImageTrainer.

    Same skeleton as the text loop, different policy constants and three differences that matter (all reproduced):
      * aggressive training ends after FIVE end-of-epoch checks in which the validation MI is below the best MI seen so far
        (`mi_not_improved == 5`, image.py:386-393) -- not at the first drop;
      * the learning rate decays (x0.5) after `decay_epoch` = 20 epochs in which the validation loss did not reach a new best
        (the comparison is against the running best loss, image.py:411, and there is no `epoch >= 15` gate), the best
        weights are reloaded and BOTH Adam optimizers are re-created (moments and step counts reset, image.py:419-420);
      * the data come from shuffled DataLoaders (a fresh order for every pass, evaluation passes included), training batches
        are dynamically binarised (torch.bernoulli, image.py:287), validation / test batches are not.
    args fields read (image.py's argparse names): kl_start, warm_up, batch_size, epochs, aggressive, nsamples, test_nepoch.

    order_fn / binarize_fn / eps_fn inject the data order, the binarisation draw and the reparameterisation noise (parity replays
    of a recorded reference run); the defaults draw them as the reference's CPU path does / on the device."""

    CLIP_GRAD, DECAY_EPOCH, LR_DECAY, MAX_DECAY, LR0 = 5.0, 20, 0.5, 5, 0.001          # image.py:18-21, 267-269

    def __init__(self, vae, x_train, x_val, x_test, args, trainer=None, log=print, np_rng=None, seed=783435, order_fn=None,
                 binarize_fn=None, eps_fn=None, epoch_hook=None, decay_epoch=None):
        from .data import ShuffledLoader
        self.vae, self.args, self.log = vae, args, log
        self.x_train = x_train
        self.train_loader = ShuffledLoader(x_train, args.batch_size, order_fn)
        self.val_loader = ShuffledLoader(x_val, args.batch_size, order_fn)
        self.test_loader = ShuffledLoader(x_test, args.batch_size, order_fn) if x_test is not None else None
        self.trainer = trainer if trainer is not None else AggressiveImageTrainer(vae, lr=self.LR0, clip=self.CLIP_GRAD, seed=seed)
        self.rng = np_rng if np_rng is not None else np.random
        self.binarize_fn, self.eps_fn, self.epoch_hook = binarize_fn, eps_fn, epoch_hook
        self.decay_epoch = self.DECAY_EPOCH if decay_epoch is None else decay_epoch
        self.kl_weight = float(args.kl_start)
        self.anneal_rate = (1.0 - args.kl_start) / (args.warm_up * len(self.train_loader))          # image.py:281
        self.opt = {"not_improved": 0, "lr": self.LR0, "best_loss": 1e4}
        self.aggressive = bool(args.aggressive)
        self.iter_ = self.decay_cnt = self.mi_not_improved = 0
        self.pre_mi = self.best_mi = 0.0
        self.best = {"loss": 1e4, "nll": 0.0, "kl": 0.0, "state": None}
        self.history, self.iterations, self.mi_checks = [], [], []

    def _binarize(self, probs):
        return self.binarize_fn(probs) if self.binarize_fn is not None else self.trainer.binarize(probs)

    def check_aggressive(self):
        """image.py:381-395, called when a full epoch worth of iterations has passed while aggressive."""
        self.vae.eval()
        with torch.no_grad():
            cur_mi = E.image_calc_mi(self.vae, self.val_loader)
        self.vae.train()
        self.mi_checks.append((self.best_mi, cur_mi))
        if cur_mi - self.best_mi < 0:
            self.mi_not_improved += 1
            if self.mi_not_improved == 5:
                self.aggressive = False
                self.log("STOP BURNING")
        else:
            self.best_mi = cur_mi
        self.pre_mi = cur_mi

    def end_of_epoch(self, epoch, loss, nll, kl):
        """Best-checkpoint and learning-rate policy (image.py:404-425).  Returns True when training should stop."""
        if loss < self.best["loss"]:
            self.log("update best loss")
            self.best.update(loss=loss, nll=nll, kl=kl, state=copy.deepcopy(self.vae.state_dict()))
        if loss > self.best["loss"]:
            self.opt["not_improved"] += 1
            if self.opt["not_improved"] >= self.decay_epoch:
                self.opt["best_loss"] = loss
                self.opt["not_improved"] = 0
                self.opt["lr"] *= self.LR_DECAY
                if self.best["state"] is not None:
                    self.vae.load_state_dict(self.best["state"])
                    self.trainer.dec.wgen += 1          # packed convolution weights are cached per weight version
                self.decay_cnt += 1
                self.log("new lr: %f" % self.opt["lr"])
                self.trainer.reset_optimizer(self.opt["lr"])
        else:
            self.opt["not_improved"] = 0
            self.opt["best_loss"] = loss
        return self.decay_cnt == self.MAX_DECAY

    def run(self):
        args, tr = self.args, self.trainer
        n_iter = len(self.train_loader)
        log_niter = max(1, n_iter // 5)                                                         # image.py:249
        start = time.time()
        self.vae.train()
        for epoch in range(args.epochs):
            if self.epoch_hook is not None:
                self.epoch_hook(self, epoch)
            rep_rec = rep_kl = 0.0
            rep_n = 0
            for probs, _ in self.train_loader:
                batch = self._binarize(probs)
                rep_n += int(batch.shape[0])
                self.kl_weight = min(1.0, self.kl_weight + self.anneal_rate)
                inner = 0
                if self.aggressive:
                    inner = tr.inner_loop(self.x_train, batch, self.kl_weight, batch_size=args.batch_size, np_rng=self.rng,
                                          eps_fn=self.eps_fn, binarize_fn=self.binarize_fn)
                tr.reset_stats()
                tr.step(batch, self.kl_weight, eps=None if self.eps_fn is None else self.eps_fn(batch),
                        update="decoder" if self.aggressive else "both")
                st = tr.read_stats()
                rep_rec += st["rec_sum"]
                rep_kl += st["kl_sum"]
                self.iterations.append(dict(epoch=epoch, iter=self.iter_, kl_weight=self.kl_weight, aggressive=self.aggressive,
                                            inner_steps=inner, rec_sum=st["rec_sum"], kl_sum=st["kl_sum"], n=int(batch.shape[0])))
                if self.iter_ % log_niter == 0:
                    train_loss = (rep_rec + rep_kl) / rep_n
                    if self.aggressive or epoch == 0:
                        self.vae.eval()
                        with torch.no_grad():
                            mi = E.image_calc_mi(self.vae, self.val_loader)
                            au, _ = E.image_calc_au(self.vae, self.val_loader)
                        self.vae.train()
                        self.log("epoch: %d, iter: %d, avg_loss: %.4f, kl: %.4f, mi: %.4f, recon: %.4f,au %d, time elapsed %.2fs" % (
                            epoch, self.iter_, train_loss, rep_kl / rep_n, mi, rep_rec / rep_n, au, time.time() - start))
                    else:
                        self.log("epoch: %d, iter: %d, avg_loss: %.4f, kl: %.4f, recon: %.4f,time elapsed %.2fs" % (
                            epoch, self.iter_, train_loss, rep_kl / rep_n, rep_rec / rep_n, time.time() - start))
                    rep_rec = rep_kl = 0.0
                    rep_n = 0
                self.iter_ += 1
                if self.aggressive and self.iter_ % n_iter == 0:
                    self.check_aggressive()
            self.log("kl weight %.4f" % self.kl_weight)
            self.vae.eval()
            with torch.no_grad():
                loss, nll, kl = E.image_test(self.vae, self.val_loader, "VAL", args, verbose=False)
                au, _ = E.image_calc_au(self.vae, self.val_loader)
            self.log("VAL --- avg_loss: %.4f, kl: %.4f, nll: %.4f, %d active units" % (loss, kl, nll, au))
            improved = loss < self.best["loss"]
            self.history.append(dict(epoch=epoch, loss=loss, nll=nll, kl=kl, au=au, aggressive=self.aggressive,
                                     kl_weight=self.kl_weight, lr=self.opt["lr"], best_updated=improved))
            stop = self.end_of_epoch(epoch, loss, nll, kl)
            self.history[-1].update(lr_after=self.opt["lr"], decay_cnt=self.decay_cnt)
            if stop:
                break
            if epoch % getattr(args, "test_nepoch", 5) == 0 and self.test_loader is not None:
                with torch.no_grad():
                    E.image_test(self.vae, self.test_loader, "TEST", args, verbose=False)
            self.vae.train()
        if self.best["state"] is not None:
            self.vae.load_state_dict(self.best["state"])
            self.trainer.dec.wgen += 1
        return dict(best_loss=self.best["loss"], best_nll=self.best["nll"], best_kl=self.best["kl"], epochs=len(self.history),
                    history=self.history)
