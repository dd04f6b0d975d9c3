"""MI355X-native counterpart of the reference's model/evaluator.py (Evaluator):
path metrics computed with an independent next-item recommender (SampleNet).

Same method names, arguments and return values as the reference.  Nothing of
shape [B, L, n_item] is built: every metric needs, per consumed row, either the
rank of one item (count epilogue), or log-softmax at one or two items
(log-sum-exp epilogue + exact gathers).  file:line = /root/reference/model/evaluator.py.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from ._backend import make_scheduler, project_ce
from .layers import get_end_index


class Evaluator(nn.Module):
    def __init__(self, config, net, device):
        super().__init__()
        self.PAD_ID = 0
        self.vocab_size = config.n_item
        self.net = net.module if isinstance(net, nn.DataParallel) else net
        self.device = device
        self.softmax = nn.LogSoftmax(dim=2)
        self.loss_function = nn.CrossEntropyLoss()
        self.optimizer = optim.Adam(filter(lambda x: x.requires_grad, self.net.parameters()), betas=(0.9, 0.98),
                                    eps=1e-09, lr=config.lr1)
        self.pla_lr_scheduler = make_scheduler(self.optimizer)

    # ---- training side (reference :53-92): decoder trunk = stock autograd (train) / HIP engine (eval); projection +
    #      cross entropy on the HIP engine in both, without the [B (L-1), n_item] logits
    def _masked_loss(self, target):
        net = self.net
        x = net.decoding(target[:, :-1])
        rows = x.reshape(-1, net.embed_dim)
        tgt = target[:, 1:].reshape(-1)
        labels0 = torch.where(tgt.gt(self.PAD_ID), tgt - 1, torch.full_like(tgt, -1)).to(torch.int64)
        if net._hip.world != 1 or rows.device.type != "cuda":  # sharded or CPU module: the reference's own formulation
            out = net.project(rows)
            mask = labels0.ge(0)
            return self.loss_function(out[mask], labels0[mask])
        return project_ce(rows, net.project, labels0, net._hip)

    def train_batch(self, target):
        self.net.train()
        loss = self._masked_loss(target)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss.item()

    def get_loss_on_eval_data(self, eval_data):
        self.net.eval()
        with torch.no_grad():
            return self._masked_loss(eval_data).item()

    # ---- helpers on the HIP path
    def _rows(self, seqs, rows_b, rows_pos):
        return self.net.decode_rows(seqs, rows_b, rows_pos)

    def _log_probs(self, xrows, ids0):
        """log softmax over the catalog at ids0 [M, g] (0-based; -1 -> 0.0)."""
        hip = self.net._hip
        mx, sm = hip.lse(xrows)
        e = hip.gather(xrows, ids0)
        lp = e.double() - mx.double().unsqueeze(1) - torch.log(sm.double()).unsqueeze(1)
        return torch.where(ids0 >= 0, lp, torch.zeros_like(lp))

    def _ranks(self, xrows, target0, excl0):
        hip = self.net._hip
        ref = hip.gather(xrows, target0.view(-1, 1))[:, 0].contiguous()
        return hip.count_before(xrows, ref, target0, excl0) + 1

    @staticmethod
    def _first_none_zero_index_batch(t):
        """Index before the first 0 of each row (reference :136-144, vectorised)."""
        L = t.size(1)
        z = (t == 0)
        first = torch.where(z.any(1), z.float().argmax(1), torch.full((t.size(0),), L, device=t.device))
        return first - 1

    def _get_first_none_zero_index(self, tensor):
        zero_list = (tensor == 0).nonzero()
        return tensor.size()[0] - 1 if len(zero_list) == 0 else zero_list[0].item() - 1

    def _get_last_path_index(self, tensor, target):
        target_pos = (tensor == target).nonzero()
        return self._get_first_none_zero_index(tensor) if len(target_pos) == 0 else target_pos[0].item() - 1

    def _delete_item_in_history(self, tensor, indices):
        return tensor[~tensor.unsqueeze(1).eq(indices).any(1)]

    @staticmethod
    def _prefix_excl(seqs, end):
        """0-based ids of seqs[b, :end[b]+1], -1 elsewhere (history filter, :267, :283)."""
        L = seqs.size(1)
        col = torch.arange(L, device=seqs.device).unsqueeze(0)
        keep = col <= end.unsqueeze(1)
        return torch.where(keep & (seqs > 0), seqs - 1, torch.full_like(seqs, -1))

    # ---- metrics
    def get_accuracy_metrics_in_batch(self, seqs, top_k=20, use_h=True):
        """(hit count, rr array) for the evaluator itself (reference :94-133)."""
        self.net.eval()
        B = seqs.size(0)
        dev = seqs.device
        end = self._first_none_zero_index_batch(seqs)  # get_end_index per row
        b = torch.arange(B, device=dev)
        label = seqs[b, end]
        xr = self._rows(seqs[:, :-1], b, end - 1)
        excl = self._prefix_excl(seqs, end - 1) if use_h else None
        ranks = self._ranks(xr, label - 1, excl).cpu().numpy()
        label_np, seqs_np, end_np = label.cpu().numpy(), seqs.cpu().numpy(), end.cpu().numpy()
        hit, rr = 0, []
        for i in range(B):
            if use_h and label_np[i] in seqs_np[i, :end_np[i]]:
                continue
            if ranks[i] <= top_k:
                hit += 1
            rr.append(np.reciprocal(float(ranks[i])))
        return hit, np.array(rr)

    def get_grad_in_batch(self, histories, new_seqs, targets, start_pos, l_paths):
        """Step-wise log-probabilities of the path items and of the target along
        the path (reference :162-243).  Returns (t_probs [B,S], p_probs [B,S],
        avg_ps list, iois list).  Unlike the reference, `histories` is not
        modified (the reference mutates it through views, :189,220-222)."""
        self.net.eval()
        dev = histories.device
        B = new_seqs.size(0)
        l_paths = l_paths.to(dev).long()
        start_pos = start_pos.to(dev).long()
        targets = targets.to(dev).long()
        S = int(l_paths.max().item())
        wins = histories[:, :-1].clone()
        Lw = wins.size(1)
        b = torch.arange(B, device=dev)
        t_probs = torch.zeros((B, S), dtype=torch.float64, device=dev)
        p_probs = torch.zeros((B, S), dtype=torch.float64, device=dev)
        col = torch.arange(Lw, device=dev).unsqueeze(0)
        for i in range(S):
            end = self._first_none_zero_index_batch(wins)
            active = i < l_paths
            idx = (start_pos + i).clamp(max=new_seqs.size(1) - 1)
            nxt = torch.where(active, new_seqs[b, idx], torch.zeros_like(targets))
            xr = self._rows(wins, b, end.clamp(min=0))
            ids0 = torch.stack([nxt - 1, targets - 1], dim=1)
            ids0 = torch.where(active.unsqueeze(1), ids0, torch.full_like(ids0, -1))
            lp = self._log_probs(xr, ids0)
            p_probs[:, i] = lp[:, 0]
            t_probs[:, i] = lp[:, 1]
            full = end == Lw - 1
            shifted = torch.cat([wins[:, 1:], nxt.unsqueeze(1)], dim=1)
            grown = torch.where(col == (end + 1).unsqueeze(1), nxt.unsqueeze(1), wins)
            wins = torch.where(full.unsqueeze(1), shifted, grown)
        t_np, p_np = t_probs.cpu().numpy(), p_probs.cpu().numpy()
        avg_ps, iois = [], []
        for i in range(B):
            tp = t_np[i][t_np[i] < 0]
            pp = p_np[i][p_np[i] < 0]
            iois.append(tp[-1] - tp[0])
            avg_ps.append(sum(pp) / len(pp))
        return t_np, p_np, avg_ps, iois

    def get_rr_increase_in_batch(self, histories, new_seqs, targets):
        """(irr, ir): change of reciprocal rank / rank of the target before and
        after the path (reference :245-290)."""
        self.net.eval()
        dev = histories.device
        B = histories.size(0)
        b = torch.arange(B, device=dev)
        targets = targets.to(dev).long()
        dec = histories[:, :-1]
        end = self._first_none_zero_index_batch(dec)
        xr = self._rows(dec, b, end.clamp(min=0))
        begin_r = self._ranks(xr, targets - 1, self._prefix_excl(dec, end)).cpu().numpy()
        dec = new_seqs[:, :-1]
        hit = dec == targets.unsqueeze(1)
        first_t = torch.where(hit.any(1), hit.float().argmax(1) - 1, self._first_none_zero_index_batch(dec))
        xr = self._rows(dec, b, first_t.clamp(min=0))
        end_r = self._ranks(xr, targets - 1, self._prefix_excl(dec, first_t)).cpu().numpy()
        irr = np.array([1 / end_r[i] - 1 / begin_r[i] for i in range(B)])
        ir = np.array([end_r[i] - begin_r[i] for i in range(B)])
        return irr, ir

    def get_pp_in_batch(self, new_seqs, start_pos, l_paths):
        """Mean negative log-likelihood of each path given its history
        (reference :292-323); list of B floats."""
        self.net.eval()
        dev = new_seqs.device
        B = new_seqs.size(0)
        start_pos = start_pos.to(dev).long()
        l_paths = l_paths.to(dev).long()
        S = int(l_paths.max().item())
        off = torch.arange(S, device=dev).unsqueeze(0)
        valid = off < l_paths.unsqueeze(1)
        pos_t = (start_pos.unsqueeze(1) + off).clamp(max=new_seqs.size(1) - 1)
        tgt = torch.gather(new_seqs, 1, pos_t)
        valid = valid & (tgt > self.PAD_ID)
        rows_b = torch.arange(B, device=dev).unsqueeze(1).expand(B, S)[valid]
        rows_pos = (pos_t - 1)[valid]
        xr = self._rows(new_seqs[:, :-1], rows_b, rows_pos.clamp(min=0))
        lp = self._log_probs(xr, (tgt[valid] - 1).view(-1, 1))[:, 0]
        nll = torch.zeros(B, dtype=torch.float64, device=dev).index_add_(0, rows_b, -lp)
        cnt = torch.zeros(B, dtype=torch.float64, device=dev).index_add_(0, rows_b, torch.ones_like(lp))
        return (nll / cnt).cpu().numpy().tolist()
