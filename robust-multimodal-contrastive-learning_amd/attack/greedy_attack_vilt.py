"""Greedy text attack on the MoCo objective - the TENSOR side of attack/greedy_attack_vilt.py:385-599
(`GreedyAttack_moco`): per loop (i) saliency = gradient of the batch InfoNCE loss wrt the word-embedding
output (get_grad :406-452), (ii) one candidate sentence per replacement, scored by the loss with that row
replaced (split_forward :454-492), (iii) keep the best candidate if it raises the loss and its index is > 0
(:562-578).  All encoder / InfoNCE work runs in librmcl_hip.so.

The LINGUISTIC side of the reference (BertTokenizer by name, nltk stop words, counter-fitted synonym
tables: greedy_attack_vilt.py:51-68,76-111) needs resources that are not available offline.
It is therefore pluggable: ``candidate_fn(loop, sample, position, ids_row) -> list[token id]`` supplies the
replacements; words are tokens.  The default is a deterministic synthetic generator with the same tensor
work as the reference (n_candidates sentences per sample and loop)."""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from .. import _lib as L

SEP_ID = 102


def synthetic_candidates(seed: int, n: int, vocab: int) -> Callable:
    def fn(loop: int, b: int, t: int, ids_row) -> List[int]:
        g = torch.Generator().manual_seed(seed * 1000003 + loop * 10007 + b * 101 + t)
        return torch.randint(1000, vocab, (n,), generator=g).tolist()
    return fn


class GreedyAttack:
    def __init__(self, config, contrastive_framework=None, candidate_fn: Optional[Callable] = None):
        self.contrastive_framework = contrastive_framework
        self.max_length = config["max_text_len"]
        self.n_candidates = config["n_candidates"]
        self.max_loops = config["max_loops"]
        self.sim_thred = config.get("sim_thred", 0.5)
        self.max_image_len = config["max_image_len"]
        self.candidate_fn = candidate_fn or synthetic_candidates(config.get("seed", 0), self.n_candidates, config["vocab_size"])

    def adv_attack_samples(self, pl_module, batch, k_modality):
        raise NotImplementedError(f"adv_attack_samples of {self.contrastive_framework} isn't implemented.")


class GreedyAttack_moco(GreedyAttack):
    def __init__(self, config, candidate_fn: Optional[Callable] = None):
        super().__init__(config, "moco", candidate_fn)

    # ---- tensor side, same method names as the reference ----------------------------------------------------------
    def get_grad(self, pl_module, pb, op, de):
        """get_grad (:406-452): forward, batch-mean InfoNCE, backward to the OUTPUT of word_embeddings (what the
        reference's backward hook captures).  Returns (per-row CE [B], grads view [B,L,D] = `de`, q [B,128])."""
        eng = pl_module.engine
        Bn = pb.B
        eng.encoder_forward(pb, key=False, mode=L.MODE_DATA, patchesT=op)
        eng.heads_forward(pb, key=False)
        eng.infonce(pb, 1.0 / Bn, want_dq=True)
        ce0 = pb.rows[:, 0].clone()
        eng.heads_backward(pb, pb.dq, None, with_grads=False)
        eng.encoder_backward(pb, L.MODE_DATA, op, pb.dcls, cls_only=True, dpatches=None, dtext=de)
        return ce0, de.view(Bn, pb.d.L, -1), pb.q

    def split_forward(self, pl_module, pc, n_real):
        """split_forward (:454-492), device part: candidates through the encoder, per-row CE against the same keys."""
        eng = pl_module.engine
        eng.encoder_forward(pc, key=False, mode=L.MODE_INFER, patchesT=pc.patchesT)
        eng.heads_forward(pc, key=False)
        eng.infonce(pc, 0.0, want_dq=False)
        return pc.rows[:n_real, 0]

    @staticmethod
    def select(ce0, cec, owner, n_real, Bn):
        """split_forward's scoring (:466-490) on per-row CE values.  The reference evaluates the BATCH-MEAN loss with row
        i replaced by candidate j and compares it with the original batch mean.  Its `t_save = ori_z[i]` (:475) is a
        view, so the restore at :489 is a no-op and row i keeps its LAST candidate while later samples are scored:
            loss_ij = mean(ce0) + sum_{r<i} (ce_{r,last} - ce0_r)/B + (ce_ij - ce0_i)/B
        Returns [(losses, best index or -1)] per sample (first maximum, strict >, like :485-486)."""
        ori = float(sum(ce0) / Bn)
        out, drift, start = [], 0.0, 0
        for b in range(Bn):
            idx = [i for i in range(start, n_real) if owner[i] == b]
            start = idx[-1] + 1
            best, best_j, losses = ori, -1, []
            for j, r in enumerate(idx):
                lj = ori + drift + (cec[r] - ce0[b]) / Bn
                losses.append(lj)
                if lj > best:
                    best, best_j = lj, j
            drift += (cec[idx[-1]] - ce0[b]) / Bn
            out.append((losses, best_j))
        return out

    def adv_attack_samples(self, pl_module, batch, k_modality):
        eng = pl_module.engine
        dev = eng.device
        ids_host = batch["text_ids"].detach().to("cpu", torch.int64).clone()
        masks = batch["text_masks"].to(dev, torch.int64)
        Bn, Lt = ids_host.shape
        nc = self.n_candidates
        Bc = Bn * nc
        pb = eng.bind_batch(ids_host.to(dev), masks, batch["image"][0], tag="txtatk")
        pc = None                                                   # candidate buffers: same image geometry, one row per candidate
        op = eng.make_operand(pb)                                   # clean image, shared by every loop
        de = torch.empty(Bn * Lt, pb.d.D, device=dev)
        k = k_modality.to(dev, torch.float32).contiguous()
        pb.k.copy_(k)
        orig = ids_host.clone()
        history = [set() for _ in range(Bn)]
        changes = [0] * Bn
        sep = [int((ids_host[b] == SEP_ID).nonzero()[0]) for b in range(Bn)]

        for loop in range(self.max_loops):
            pb.text_ids = ids_host.to(dev)
            ce0, grads, _ = self.get_grad(pl_module, pb, op, de)
            sal = grads.abs().sum(-1).cpu()                          # L1 norm of the gradient per position (:221-228)
            # ---- pick one position per sample, build the candidate sentences (host logic) -------------
            rows, owner, pos_of = [], [], []
            for b in range(Bn):
                max_len = int(sep[b] * 0.2)
                order = torch.argsort(sal[b, 1:sep[b]], descending=True, stable=True) + 1
                chosen = None
                for t in order.tolist():
                    if t in history[b] or changes[b] >= min(max_len, self.max_loops):
                        continue
                    chosen = t
                    break
                if chosen is None:
                    rows.append(ids_host[b].clone()); owner.append(b); pos_of.append(None)
                    continue
                history[b].add(chosen)
                for c in self.candidate_fn(loop, b, chosen, ids_host[b])[:nc]:
                    r = ids_host[b].clone(); r[chosen] = c
                    rows.append(r); owner.append(b); pos_of.append(chosen)
            n_real = len(rows)
            while len(rows) < Bc:                                      # pad to the fixed candidate batch
                rows.append(ids_host[0].clone()); owner.append(0)
            own = torch.tensor(owner, device=dev)
            pc = eng.twin(pb, "txtatk_cand", owner=own)
            pc.text_ids = torch.stack(rows).to(dev)
            pc.text_mask = masks.index_select(0, own)
            torch.index_select(op.view(Bn, -1), 0, own, out=pc.patchesT.view(Bc, -1))
            pc.k.copy_(k.index_select(0, own))
            cec = self.split_forward(pl_module, pc, n_real).cpu().tolist()
            picks = self.select(ce0.cpu().tolist(), cec, owner, n_real, Bn)
            # ---- selection (:562-578): a changed sample takes its best candidate when its index is > 0 ---
            start = 0
            for b, (_, best_j) in enumerate(picks):
                first = start
                start += len(picks[b][0])
                if pos_of[first] is None:
                    continue
                if best_j > 0:
                    changes[b] += 1
                    ids_host[b] = rows[first + best_j]

        nchg = [(orig[b] != ids_host[b]).sum().item() for b in range(Bn)]
        nwords = [max(sep[b] - 1, 1) for b in range(Bn)]
        return {"txt_input_ids": ids_host.to(dev), "text_masks": masks, "text": batch.get("text"),
                "num_changes": sum(nchg) / Bn, "change_rate": sum(c / n for c, n in zip(nchg, nwords)) / Bn,
                "Problem": any(c == 0 for c in nchg), "changes_verification": changes}
