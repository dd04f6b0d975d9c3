"""The callers either side of the hot path: the evaluation loops of the
reference's `pipeline.py`, restated so that a run on the GPU box needs nothing from the reference.

    test_model      reference pipeline.py:151-247   PIF -> ranking metrics -> 20-step path, per batch;
                                                     Hit / MRR / early-success aggregates; the four result
                                                     arrays paths_d_<n>.npy, histories.npy, targets_d_<n>.npy, r_u.npy
    evaluate_prob   reference pipeline.py:250-331   perplexity, rank increase, step-wise log-probs of the
                                                     saved paths under the evaluator; p_probs.npy, t_probs.npy

Same loop order, same aggregation arithmetic (Python float sums over numpy scalars), same printed lines,
same file names, dtypes and shapes.  The data-side helpers restate the two loaders the loops consume:
`DataLoaderEvalIRS` (data_provider.py:577-617) and `DatasetEvalNN1` / `DataLoaderEvalNN1`
(data_provider.py:711-783).  The handlers (`IRSNN`, `Evaluator`) are the drop-in classes of
`influentialrs_amd.model`; unlike the published IRN they accept any batch size.
"""
import os
from typing import Iterable, List, Sequence

import numpy as np
import torch

from . import synth
from .model.layers import get_item_index


# ------------------------------------------------------------------ loaders
def eval_batches_irs(rows: Sequence, batch_size: int, seq_len: int, gap_len: int):
    """DataLoaderEvalIRS(shuffle=False) over rows [new_seq, user_id, random_target, label]:
    yields (raw list[LongTensor], seq LongTensor[B, L], users, targets, labels)."""
    for i in range(0, len(rows), batch_size):
        chunk = rows[i:i + batch_size]
        raws, seqs, users, targets, labels = synth.collate_eval_irs(chunk, seq_len, gap_len=gap_len)
        yield ([torch.from_numpy(np.asarray(r, dtype=np.int64)) for r in raws], torch.from_numpy(seqs),
               torch.from_numpy(users), torch.from_numpy(targets), torch.from_numpy(labels))


def build_eval_nn1(histories: Iterable, paths: np.ndarray, targets: Sequence[int], seq_len: int):
    """DatasetEvalNN1._preprocess_seqs (data_provider.py:716-754): per row
    [pad_history (L+1), new_seq (L+1), target, start_pos, l_path]."""
    data = []
    l_seq = seq_len + 1
    for history, path, target in zip(histories, paths, targets):
        history = np.asarray(history)
        path = np.asarray(path)
        path = path[path > 0]
        if get_item_index(path, target) == -1:  # not an early success: the target closes the path
            path = np.concatenate((path, [target]))
        l_path = len(path)
        new_seq = np.zeros(l_seq)
        actual = np.concatenate((history, path))
        if len(actual) <= l_seq:
            new_seq[:len(actual)] = actual
            start_pos = len(history)
        else:
            new_seq[:] = actual[-l_seq:]
            start_pos = l_seq - len(path)
        pad_history = np.zeros(l_seq)
        if len(history) <= seq_len:
            pad_history[:len(history)] = history
        else:
            pad_history[:-1] = history[-seq_len:]
        data.append([pad_history, new_seq, target, start_pos, l_path])
    return data


def eval_batches_nn1(data: List, batch_size: int):
    """DataLoaderEvalNN1(shuffle=False): (histories, new_seqs, targets, start_pos, l_path) tensors."""
    for i in range(0, len(data), batch_size):
        chunk = data[i:i + batch_size]
        yield (torch.LongTensor(np.array([r[0] for r in chunk])), torch.LongTensor(np.array([r[1] for r in chunk])),
               torch.tensor([r[2] for r in chunk]), torch.tensor([r[3] for r in chunk]), torch.tensor([r[4] for r in chunk]))


def _save_ragged(path: str, seqs: List[np.ndarray]):
    """histories.npy as the reference's numpy wrote it: a 1-D object array of int64 arrays."""
    o = np.empty(len(seqs), dtype=object)
    for i, x in enumerate(seqs):
        o[i] = x
    np.save(path, o)


# ------------------------------------------------------------------ pipeline.test_model
def test_model(config, rows: Sequence, irn, device, result_dir: str = None, verbose: bool = True):
    """The random-target evaluation of reference pipeline.py:151-247 over `rows` (what
    dp.get_random_evaluate_data returned there).  Returns a dict with the aggregates and the arrays;
    writes the four .npy files when `result_dir` is given."""
    say = print if verbose else (lambda *a, **k: None)
    n_eval = len(rows)
    with torch.no_grad():
        irn.eval()
        paths, histories, targets, reverse_ranks, r_us = [], [], [], [], []
        n_early_success, hit = 0, 0
        for raw, seq, u, t, l in eval_batches_irs(rows, config.batch_size, config.max_len, config.gap_len):
            t, l, u, seq = t.to(device), l.to(device), u.to(device), seq.to(device)
            r_u = irn.get_pif_in_batch(seq, u)
            r_us = r_u if len(r_us) == 0 else np.concatenate((r_us, r_u))
            hit_count, rr = irn.get_accuracy_metrics_in_batch(raw, seq, u, t, l, config.top_k, config.gap_len, config.use_h)
            hit += hit_count
            reverse_ranks = rr if len(reverse_ranks) == 0 else np.concatenate([reverse_ranks, rr])
            p, t_out, h, early_success = irn.get_seq_in_batch(seq, u, t, config.max_path_len, config.gap_len, config.sample,
                                                              config.sample_k)
            n_early_success += early_success
            if len(paths) == 0:
                paths, histories, targets = p, h, t_out
            else:
                paths = np.vstack((paths, p))
                histories += h
                targets = np.concatenate((targets, t_out))
        hit = hit / n_eval
        mrr = sum(reverse_ranks) / n_eval
        say("Hit rate: %f" % hit)
        say("MRR: %f" % mrr)
        say("Early Success: %i" % n_early_success)
        say("Early Success Rate: %f" % (n_early_success / n_eval))
    if result_dir is not None:
        os.makedirs(result_dir, exist_ok=True)
        np.save(os.path.join(result_dir, "paths_d_%i.npy" % config.max_path_len), paths)
        _save_ragged(os.path.join(result_dir, "histories.npy"), histories)
        np.save(os.path.join(result_dir, "targets_d_%i.npy" % config.max_path_len), targets)
        np.save(os.path.join(result_dir, "r_u.npy"), r_us)
    return dict(hit=hit, mrr=mrr, n_early_success=n_early_success, early_success_rate=n_early_success / n_eval,
                paths=paths, histories=histories, targets=targets, r_u=r_us)


# ------------------------------------------------------------------ pipeline.evaluate_prob
def evaluate_prob(config, eval_config, evaluator, device, result_dir: str = None, results: dict = None,
                  verbose: bool = True):
    """Probability metrics of generated paths (reference pipeline.py:250-331).  The paths come from
    `results` (what test_model returned) or from the .npy files under `result_dir`."""
    say = print if verbose else (lambda *a, **k: None)
    if results is None:
        histories = np.load(os.path.join(result_dir, "histories.npy"), allow_pickle=True)
        targets = np.load(os.path.join(result_dir, "targets_d_%i.npy" % config.max_path_len), allow_pickle=True)
        paths = np.load(os.path.join(result_dir, "paths_d_%i.npy" % config.max_path_len), allow_pickle=True)
    else:
        histories, targets, paths = results["histories"], results["targets"], results["paths"]
    data = build_eval_nn1(histories, paths, targets, seq_len=eval_config.max_len)
    with torch.no_grad():
        evaluator.eval()
        perplexity, p_probs, t_probs, iis, irrs, irs, avg_acpt = [], [], [], [], [], [], []
        for h, d, t, sp, lp in eval_batches_nn1(data, eval_config.batch_size):
            h, d, t, sp, lp = h.to(device), d.to(device), t.to(device), sp.to(device), lp.to(device)
            pp = evaluator.get_pp_in_batch(d, sp, lp)
            irr, ir = evaluator.get_rr_increase_in_batch(h, d, t)
            target_ps, path_ps, avg_ps, ii = evaluator.get_grad_in_batch(h, d, t, sp, lp)
            if len(perplexity) == 0:
                perplexity, iis, irrs, irs, avg_acpt, p_probs, t_probs = pp, ii, irr, ir, avg_ps, path_ps, target_ps
            else:
                perplexity = np.concatenate((perplexity, pp))
                iis = np.concatenate((iis, ii))
                irrs = np.concatenate((irrs, irr))
                irs = np.concatenate((irs, ir))
                avg_acpt = np.concatenate((avg_acpt, avg_ps))
                p_probs = np.vstack((p_probs, path_ps))
                t_probs = np.vstack((t_probs, target_ps))
        length = len(perplexity)
        out = dict(perplexity=sum(perplexity) / length, ii=sum(iis) / length, irr=sum(irrs) / length,
                   ir=sum(irs) / length, ap=sum(avg_acpt) / length, p_probs=p_probs, t_probs=t_probs)
        say("The perplexity: %f" % out["perplexity"])
        say("The increase of interest: %f" % out["ii"])
        say("The increase of reverse ranking: %f" % out["irr"])
        say("The increase of ranking: %f" % out["ir"])
        say("The average acceptance probability: %f" % out["ap"])
    if result_dir is not None:
        os.makedirs(result_dir, exist_ok=True)
        np.save(os.path.join(result_dir, "p_probs.npy"), p_probs)
        np.save(os.path.join(result_dir, "t_probs.npy"), t_probs)
    return out
