"""Evaluation loops of the reference's text.py on top of the HIP forward.

`test` (text.py:120-160), `calc_mi` (text.py:186-198), `calc_au` (text.py:200-227) and `calc_iwnll` (text.py:162-184) with
the reference's signatures and return values; the per-batch statistics run in lv_eval.hip kernels and stay on the device
(one host read per reported number instead of one `.item()` per batch).
"""
import math

import numpy as np
import torch

from . import engine as _eng


def test(model, test_data_batch, mode, args=None, verbose=True, np_rng=None):
    """text.py:120-160: report loss / KL / reconstruction NLL / perplexity over the batches (eval mode is the caller's).
    Returns (test_loss, nll, kl, ppl, mutual_info)."""
    report_kl_loss = report_rec_loss = 0.0
    report_num_words = report_num_sents = 0
    order = (np_rng if np_rng is not None else np.random).permutation(len(test_data_batch))     # text.py:127
    kls, recs = [], []
    for i in order:
        batch_data = test_data_batch[i]
        batch_size, sent_len = batch_data.size()
        report_num_words += (sent_len - 1) * batch_size           # not predicting the start symbol
        report_num_sents += batch_size
        nsamples = getattr(args, "nsamples", 1) if args is not None else 1
        loss, loss_rc, loss_kl = model.loss(batch_data, 1.0, nsamples=nsamples)
        recs.append(loss_rc.sum())
        kls.append(loss_kl.sum())
    report_rec_loss = float(torch.stack(recs).sum().item())
    report_kl_loss = float(torch.stack(kls).sum().item())
    mutual_info = calc_mi(model, test_data_batch)
    test_loss = (report_rec_loss + report_kl_loss) / report_num_sents
    nll = (report_kl_loss + report_rec_loss) / report_num_sents
    kl = report_kl_loss / report_num_sents
    ppl = math.exp(nll * report_num_sents / report_num_words)
    if verbose:
        print("%s --- avg_loss: %.4f, kl: %.4f, mi: %.4f, recon: %.4f, nll: %.4f, ppl: %.4f" %
              (mode, test_loss, kl, mutual_info, report_rec_loss / report_num_sents, nll, ppl))
    return test_loss, nll, kl, ppl, mutual_info


def calc_iwnll(model, test_data_batch, args, ns=100, verbose=False, np_rng=None):
    """text.py:162-184: importance-weighted NLL and perplexity over the batches (visited in a random order, as the reference
    does: the order decides which Gaussian draws each batch sees).  Returns (nll, ppl)."""
    tot = []
    report_num_words = report_num_sents = 0
    for i in (np_rng if np_rng is not None else np.random).permutation(len(test_data_batch)):
        batch_data = test_data_batch[i]
        batch_size, sent_len = batch_data.size()
        report_num_words += (sent_len - 1) * batch_size
        report_num_sents += batch_size
        tot.append(model.nll_iw(batch_data, nsamples=args.iw_nsamples, ns=ns).sum())
    report_nll_loss = float(torch.stack(tot).sum().item())
    nll = report_nll_loss / report_num_sents
    ppl = math.exp(nll * report_num_sents / report_num_words)
    if verbose:
        print("iw nll: %.4f, iw ppl: %.4f" % (nll, ppl))
    return nll, ppl


def calc_mi(model, test_data_batch):
    """text.py:186-198: batch-size-weighted mean of the per-batch mutual information estimates."""
    mi = 0.0
    num_examples = 0
    for batch_data in test_data_batch:
        batch_size = batch_data.size(0)
        num_examples += batch_size
        mi += model.calc_mi_q(batch_data) * batch_size
    return mi / num_examples


def calc_au(model, test_data_batch, delta=0.01):
    """text.py:200-227: number of active units = latent dimensions whose posterior mean varies (variance over the data
    >= delta).  Two passes over the data; the per-dimension sums accumulate on the device (lv_au_accum_f32).
    Returns (count, au_var)."""
    acc = None
    cnt = 0
    for batch_data in test_data_batch:
        mean, _ = model.encode_stats(batch_data)
        if acc is None:
            acc = torch.zeros(mean.shape[1], dtype=torch.float32, device=mean.device)
        _eng.au_accumulate(mean.detach(), None, acc)
        cnt += mean.size(0)
    mean_mean = (acc / cnt).contiguous()
    var = torch.zeros_like(acc)
    cnt = 0
    for batch_data in test_data_batch:
        mean, _ = model.encode_stats(batch_data)
        _eng.au_accumulate(mean.detach(), mean_mean, var)
        cnt += mean.size(0)
    au_var = var / (cnt - 1)
    return int((au_var >= delta).sum().item()), au_var


# ---- image side (reference image.py:96-187): the loaders yield (batch, label) pairs like DataLoader(TensorDataset(x, y)) --------
def image_test(model, test_loader, mode, args=None, verbose=True):
    """image.py:96-128: loss / KL / reconstruction over one pass of the loader (eval mode is the caller's), then calc_mi over a
    second pass.  Returns (test_loss, nll, kl)."""
    kls, recs = [], []
    report_num_examples = 0
    nsamples = getattr(args, "nsamples", 1) if args is not None else 1
    for batch_data, _ in test_loader:
        report_num_examples += batch_data.size(0)
        loss, loss_rc, loss_kl = model.loss(batch_data, 1.0, nsamples=nsamples)
        recs.append(loss_rc.sum())
        kls.append(loss_kl.sum())
    report_rec_loss = float(torch.stack(recs).sum().item())
    report_kl_loss = float(torch.stack(kls).sum().item())
    mutual_info = image_calc_mi(model, test_loader)
    test_loss = (report_rec_loss + report_kl_loss) / report_num_examples
    nll = (report_kl_loss + report_rec_loss) / report_num_examples
    kl = report_kl_loss / report_num_examples
    if verbose:
        print("%s --- avg_loss: %.4f, kl: %.4f, mi: %.4f, recon: %.4f, nll: %.4f" %
              (mode, test_loss, kl, mutual_info, report_rec_loss / report_num_examples, nll))
    return test_loss, nll, kl


def image_calc_mi(model, test_loader):
    """image.py:130-140."""
    mi = 0.0
    num_examples = 0
    for batch_data, _ in test_loader:
        batch_size = batch_data.size(0)
        num_examples += batch_size
        mi += model.calc_mi_q(batch_data) * batch_size
    return mi / num_examples


def image_calc_au(model, test_loader, delta=0.01):
    """image.py:142-162: ONE pass over the loader (the posterior means are collected, then their variance over the data is
    taken), active = variance >= delta.  Returns (count, au_var)."""
    means = [model.encode_stats(batch_data)[0].detach() for batch_data, _ in test_loader]
    means = torch.cat(means, dim=0).contiguous()
    ns = means.size(0)
    acc = torch.zeros(means.shape[1], dtype=torch.float32, device=means.device)
    _eng.au_accumulate(means, None, acc)
    mean_mean = (acc / ns).contiguous()
    var = torch.zeros_like(acc)
    _eng.au_accumulate(means, mean_mean, var)
    au_var = var / (ns - 1)
    return int((au_var >= delta).sum().item()), au_var


def image_calc_iwnll(model, test_loader, args, verbose=False):
    """image.py:164-187: importance-weighted NLL per example."""
    tot = []
    report_num_examples = 0
    for batch_data, _ in test_loader:
        report_num_examples += batch_data.size(0)
        tot.append(model.nll_iw(batch_data, nsamples=args.iw_nsamples).sum())
    nll = float(torch.stack(tot).sum().item()) / report_num_examples
    if verbose:
        print("iw nll: %.4f" % nll)
    return nll
