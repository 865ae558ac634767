"""Greedy text attack on the MoCo objective (attack/greedy_attack_vilt.py:47-599, `GreedyAttack` / `GreedyAttack_moco`):
per loop (i) saliency = gradient of the batch InfoNCE loss wrt the word-embedding output (get_grad :406-452), (ii) one
candidate sentence per replacement, scored by the loss with that row replaced (split_forward :454-492), (iii) keep the
best candidate if it raises the loss and its index is > 0 (:562-578).  All encoder / InfoNCE work runs in librmcl_hip.so.

Two front ends share that tensor side:

* WORD level = the reference's algorithm (row f4): words decoded from the ids, word <-> sub-word map, importance = L1 norm
  of the mean sub-word gradient, stop-word / history / 20 % filters, synonyms from the counter-fitted table, candidates
  RE-TOKENISED (ids and masks change length).  Needs a tokenizer (object or LOCAL vocab file) and, for
  ``synonym="cos_sim"``, the vector file at ``config["embedding_path"]``: attack/word_substitution.py.  The reference
  loads both by name from the network (greedy_attack_vilt.py:53,66-67); offline they are injected.
* TOKEN level (no tokenizer available): ``candidate_fn(loop, sample, position, ids_row) -> list[token id]`` supplies the
  replacements; words are tokens.  The default is a deterministic synthetic generator with the same tensor work as the
  reference (n_candidates sentences per sample and loop) - what bench.py's full_rmcl configuration runs."""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import numpy as np
import torch

from .. import _lib as L
from . import word_substitution as WS

SEP_ID = 102


def synthetic_candidates(seed: int, n: int, vocab: int) -> Callable:
    def fn(loop: int, b: int, t: int, ids_row) -> List[int]:
        g = torch.Generator().manual_seed(seed * 1000003 + loop * 10007 + b * 101 + t)
        return torch.randint(1000, vocab, (n,), generator=g).tolist()
    return fn


class GreedyAttack:
    def __init__(self, config, contrastive_framework=None, candidate_fn: Optional[Callable] = None, tokenizer=None,
                 stopwords=None, synonyms=None):
        """tokenizer: object, or None = ``config["tokenizer"]`` when that is a local vocab path (a hub name cannot be fetched:
        token-level mode).  stopwords: iterable / file (the reference: nltk's English list, :51).  synonyms: a ready
        ``word -> candidates`` callable with ``in`` support, else built from ``config["embedding_path"]`` (:66-67)."""
        self.contrastive_framework = contrastive_framework
        self.max_length = config["max_text_len"]
        self.n_candidates = config["n_candidates"]
        self.max_loops = config["max_loops"]
        self.sim_thred = config.get("sim_thred", 0.5)
        self.max_image_len = config["max_image_len"]
        self.synonym = config.get("synonym", "cos_sim")
        self.candidate_fn = candidate_fn or synthetic_candidates(config.get("seed", 0), self.n_candidates, config["vocab_size"])
        self.tokenizer = WS.load_tokenizer(tokenizer if tokenizer is not None else config.get("tokenizer"))
        self.check_word = WS.WordFilter(WS.load_stopwords(stopwords if stopwords is not None else config.get("stopwords")))
        self.synonyms = synonyms
        self.words_to_sub_words: List[dict] = []
        self.replace_history: List[set] = []
        self.changes_verification: List[int] = []
        if self.tokenizer is not None and self.synonyms is None:
            if self.synonym == "cos_sim" and config.get("cos_sim", True):
                path = config.get("embedding_path")
                if not path or not os.path.isfile(path):
                    raise FileNotFoundError(f"text attack: counter-fitted vectors not found at embedding_path={path!r} "
                                            "(greedy_attack_vilt.py:66-67; pass synonyms= or a local file)")
                self.init_matrix(path, config.get("sim_path"))
            elif self.synonym == "synonym":
                raise RuntimeError("text attack: synonym='synonym' needs nltk's WordNet (greedy_attack_vilt.py:205-219), which "
                                   "is not installed; pass synonyms=<callable> or use synonym='cos_sim'")
            else:
                raise ValueError("Only use wordnet of cos sim to find new words!")

    # ---- linguistic side, same method names as the reference -------------------------------------------------------
    @property
    def sim_word2id(self):
        return getattr(self.synonyms, "word2id", None)

    def init_matrix(self, embedding_path, sim_path=None):
        self.synonyms = WS.SynonymTable(embedding_path, self.n_candidates, self.sim_thred, sim_path)

    def get_synonym_by_cos(self, word):
        return list(self.synonyms(word))

    def get_important_scores(self, grads, words_to_sub_words):
        return WS.importance_scores(grads, words_to_sub_words)

    def get_inputs(self, sentences, tokenizer=None, device=None):
        ids, masks = WS.encode_sentences(tokenizer or self.tokenizer, sentences, self.max_length)
        return (ids, masks) if device is None else (ids.to(device), masks.to(device))

    def calc_words_to_sub_words(self, words, batch_size):
        self.words_to_sub_words = [WS.words_to_sub_words(self.tokenizer, words[i], self.max_length) for i in range(batch_size)]

    def compute_word_importance(self, words, input_ids, grads, batch_size):
        """compute_word_importance (:266-310) behind get_grad: per sentence the attackable word with the largest importance
        (None when every word is filtered, already replaced, or the 20 % / max_loops budget is spent).
        grads [B, L, D] host array of the saliency gradients, input_ids [B, L] host tensor."""
        sep_id = self.tokenizer.convert_tokens_to_ids("[SEP]")
        sep_idx = (input_ids == sep_id).nonzero()
        assert len(sep_idx) == batch_size
        known = self.synonyms if hasattr(self.synonyms, "__contains__") else None
        replace_idx = []
        for i in range(batch_size):
            norms = self.get_important_scores(grads[i][1:], self.words_to_sub_words[i])     # [1:]: the map skips [CLS]
            order = torch.topk(torch.tensor(norms), k=len(norms)).indices
            budget = min(int(sep_idx[i][1] * 0.2), self.max_loops)                            # at most 20 % of the words
            pick = None
            for idx in order.tolist():
                word = words[i][idx].strip().lower()
                if self.check_word(word) or (known is not None and word not in known):
                    continue
                if idx in self.replace_history[i] or self.changes_verification[i] >= budget:
                    continue
                pick = idx
                break
            replace_idx.append(pick)
            if pick is not None:
                self.replace_history[i].add(pick)
        return replace_idx

    def construct_new_samples(self, word_idx, words, batch_size):
        """construct_new_samples (:312-344): one sentence per synonym of the chosen word (the sentence itself when no word
        was chosen).  Returns (sentences, count per sample, changed flag per sample)."""
        if self.synonym not in ("cos_sim", "synonym"):
            raise ValueError("Only use wordnet of cos sim to find new words!")
        all_new_text, all_num, changed = [], [], []
        for i in range(batch_size):
            if word_idx[i] is None:
                all_new_text.append(" ".join(words[i]))
                all_num.append(1)
                changed.append(False)
                continue
            cands = self.get_synonym_by_cos(words[i][word_idx[i]])
            if self.synonym == "synonym":
                cands = cands[:self.n_candidates]
            for new_word in cands:
                sent = list(words[i])
                sent[word_idx[i]] = new_word
                all_new_text.append(" ".join(sent))
            all_num.append(len(cands))
            changed.append(True)
        return all_new_text, all_num, changed

    # ---- framework hooks (get_grad / split_forward of the reference's subclasses) -----------------------------------
    def bind_keys(self, pl_module, pb, k):
        raise NotImplementedError(f"adv_attack_samples of {self.contrastive_framework} isn't implemented.")

    def bind_candidate_keys(self, pc, k, own):
        pass

    def get_grad(self, pl_module, pb, op, de):
        raise NotImplementedError(f"get_grad of {self.contrastive_framework} isn't implemented.")

    def score(self, pl_module, pc, ctx, owner, n_real, Bn):
        """split_forward: [(candidate losses, index of the best candidate or -1)] per sample"""
        raise NotImplementedError(f"split_forward of {self.contrastive_framework} isn't implemented.")

    def adv_attack_samples(self, pl_module, batch, k_modality):
        if self.tokenizer is not None:
            return self._attack_words(pl_module, batch, k_modality)
        return self._attack_tokens(pl_module, batch, k_modality)

    def _attack_words(self, pl_module, batch, k_modality):
        """adv_attack_samples (:494-599), word level.  Per loop: saliency on the current sentences, one word per sentence,
        its synonyms as re-tokenised candidate sentences (ids AND masks of a candidate may differ in length from its
        sentence), batch-mean CE with row i replaced, keep the best candidate when its index is > 0 (:568; index 0 is never
        taken - reference behaviour), re-tokenise the batch."""
        eng = pl_module.engine
        dev = eng.device
        tok = self.tokenizer
        ids_host = batch["text_ids"].detach().to("cpu", torch.int64).clone()
        masks_host = batch["text_masks"].detach().to("cpu", torch.int64).clone()
        Bn, Lt = ids_host.shape
        Bc = Bn * self.n_candidates
        original_words = [WS.decode_words(tok, ids_host[b]) for b in range(Bn)]
        cur_words = [list(w) for w in original_words]
        self.calc_words_to_sub_words(cur_words, Bn)
        self.replace_history = [set() for _ in range(Bn)]
        self.changes_verification = [0] * Bn
        pb = eng.bind_batch(ids_host.to(dev), masks_host.to(dev), batch["image"][0], tag="txtatk")
        op = eng.make_operand(pb)                                   # clean image, shared by every loop
        de = torch.empty(Bn * Lt, pb.d.D, device=dev)
        k = k_modality.to(dev, torch.float32).contiguous()
        self.bind_keys(pl_module, pb, k)
        text = [" ".join(w) for w in cur_words]
        self.trace = []                                             # per loop: (replace_idx, all_new_text, all_num, picks) for tests

        for loop in range(self.max_loops):
            pb.text_ids = ids_host.to(dev)
            pb.text_mask = masks_host.to(dev)
            ctx, grads, _ = self.get_grad(pl_module, pb, op, de)
            replace_idx = self.compute_word_importance(cur_words, ids_host, grads.cpu().numpy(), Bn)
            all_new_text, all_num, changed = self.construct_new_samples(replace_idx, cur_words, Bn)
            n_real = len(all_new_text)
            if n_real > Bc:
                raise RuntimeError(f"text attack: {n_real} candidate sentences exceed batch x n_candidates = {Bc}")
            cids, cmasks = self.get_inputs(all_new_text)
            owner = [b for b in range(Bn) for _ in range(all_num[b])]
            pad = Bc - n_real
            own = torch.tensor(owner + [0] * pad, device=dev)
            pc = eng.twin(pb, "txtatk_cand", owner=own)
            pc.text_ids = torch.cat([cids, ids_host[:1].expand(pad, Lt)]).to(dev).contiguous()
            pc.text_mask = torch.cat([cmasks, masks_host[:1].expand(pad, Lt)]).to(dev).contiguous()
            torch.index_select(op.view(Bn, -1), 0, own, out=pc.patchesT.view(Bc, -1))
            self.bind_candidate_keys(pc, k, own)
            picks = self.score(pl_module, pc, ctx, owner, n_real, Bn)
            count = 0
            for b, (losses, best_j) in enumerate(picks):
                if changed[b] and best_j > 0:
                    self.changes_verification[b] += 1
                    cur_words[b] = all_new_text[best_j + count].split(" ")
                    self.words_to_sub_words[b] = WS.words_to_sub_words(tok, cur_words[b], self.max_length)
                count += len(losses)
            self.trace.append((list(replace_idx), list(all_new_text), list(all_num), [j for _, j in picks]))
            text = [" ".join(w) for w in cur_words]
            ids_host, masks_host = self.get_inputs(text)

        num_changes, change_rate = [], []
        for old, new in zip(original_words, cur_words):
            n = int(np.sum(~(np.array(old) == np.array(new))))
            num_changes.append(n)
            change_rate.append(n / len(old))
        return {"txt_input_ids": ids_host.to(dev), "text_masks": masks_host.to(dev), "text": text,
                "num_changes": float(np.mean(num_changes)), "change_rate": float(np.mean(change_rate)),
                "Problem": any(n == 0 for n in num_changes), "changes_verification": self.changes_verification}

    def _attack_tokens(self, pl_module, batch, k_modality):
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
        self.bind_keys(pl_module, pb, k)
        orig = ids_host.clone()
        history = [set() for _ in range(Bn)]
        changes = [0] * Bn
        sep = [int((ids_host[b] == SEP_ID).nonzero()[0]) for b in range(Bn)]

        for loop in range(self.max_loops):
            pb.text_ids = ids_host.to(dev)
            ctx, grads, _ = self.get_grad(pl_module, pb, op, de)
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
            self.bind_candidate_keys(pc, k, own)
            picks = self.score(pl_module, pc, ctx, owner, n_real, Bn)
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


class GreedyAttack_moco(GreedyAttack):
    def __init__(self, config, candidate_fn: Optional[Callable] = None, tokenizer=None, stopwords=None, synonyms=None):
        super().__init__(config, "moco", candidate_fn, tokenizer, stopwords, synonyms)

    # ---- tensor side, same method names as the reference ----------------------------------------------------------
    def get_grad(self, pl_module, pb, op, de):
        """get_grad (:406-452): forward, batch-mean InfoNCE, backward to the OUTPUT of word_embeddings (what the
        reference's backward hook captures).  Returns (per-row CE [B], grads view [B,L,D] = `de`, q [B,128])."""
        eng = pl_module.engine
        Bn = pb.B
        eng.encoder_forward(pb, key=False, mode=L.MODE_DATA, patchesT=op, cls_tail=True)
        eng.heads_forward(pb, key=False)
        eng.infonce(pb, 1.0 / Bn, want_dq=True)
        ce0 = pb.rows[:, 0].clone()
        eng.heads_backward(pb, pb.dq, None, with_grads=False)
        eng.encoder_backward(pb, L.MODE_DATA, op, pb.dcls, cls_only=True, dpatches=None, dtext=de)
        return ce0, de.view(Bn, pb.d.L, -1), pb.q

    def split_forward(self, pl_module, pc, n_real):
        """split_forward (:454-492), device part: candidates through the encoder, per-row CE against the same keys."""
        eng = pl_module.engine
        eng.encoder_forward(pc, key=False, mode=L.MODE_INFER, patchesT=pc.patchesT, cls_tail=True)
        eng.heads_forward(pc, key=False)
        eng.infonce(pc, 0.0, want_dq=False)
        return pc.rows[:n_real, 0]

    def bind_keys(self, pl_module, pb, k):
        pb.k.copy_(k)

    def bind_candidate_keys(self, pc, k, own):
        pc.k.copy_(k.index_select(0, own))

    def score(self, pl_module, pc, ctx, owner, n_real, Bn):
        cec = self.split_forward(pl_module, pc, n_real).cpu().tolist()
        return self.select(ctx.cpu().tolist(), cec, owner, n_real, Bn)

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


class GreedyAttack_barlowtwins(GreedyAttack):
    """attack/greedy_attack_vilt.py:602-700: the same greedy attack maximising the Barlow-Twins loss
    on_diag + adv_lr * off_diag of c = q^T k / B (local batch; q = barlowtwins_head(cls_feats): the reference attacks a deep
    copy of the head, which keeps the module's mode - in training BATCH statistics, so a candidate batch is normalised by its
    own rows; in validation the running estimates)."""

    def __init__(self, config, candidate_fn: Optional[Callable] = None, tokenizer=None, stopwords=None, synonyms=None):
        super().__init__(config, "barlowtwins", candidate_fn, tokenizer, stopwords, synonyms)
        self._zk = None

    def bind_keys(self, pl_module, pb, k):
        self._zk = k

    def get_grad(self, pl_module, pb, op, de):
        """get_grad (:623-668): loss = on_diag + adv_lr * off_diag, gradient at the output of word_embeddings.
        Returns (context for score(): the projections [B, H3], grads view [B,L,D], the projections)."""
        eng = pl_module.engine
        bb = eng.bt_bufs(pb.B, "txtatk")
        eng.encoder_forward(pb, key=False, mode=L.MODE_DATA, patchesT=op, cls_tail=True)
        eng.heads_forward(pb, key=False, want_q=False)
        mode = bool(pl_module.training)           # the attacked head is a deep copy: it keeps the module's train / eval flag
        eng.bt_forward(bb, pb.cls, training=mode, track=False)
        eng.bt_loss(bb, self._zk, float(pb.B), pl_module.adv_lr, 1.0, want_dz=True)
        dcls = eng.bt_backward(bb, bb.dz, training=mode, with_grads=False)
        eng.heads_backward(pb, None, dcls, with_grads=False)
        eng.encoder_backward(pb, L.MODE_DATA, op, pb.dcls, cls_only=True, dpatches=None, dtext=de)
        z = bb.z.clone()
        return z, de.view(pb.B, pb.d.L, -1), z

    def score(self, pl_module, pc, ctx, owner, n_real, Bn):
        """split_forward (:670-707).  Candidates go through encoder + head as ONE batch of n_real rows (BatchNorm statistics
        over exactly those rows); then, sample by sample and candidate by candidate, row i of the projection matrix is
        replaced and the loss of the whole matrix re-evaluated.  `t_save = ori_z[i]` (:691) is a view, so row i keeps its LAST
        candidate while later samples are scored (same reference behaviour as the MoCo attack); the comparison baseline is
        the loss of the UNMODIFIED matrix for every sample (:684-688).  All n_real + 1 losses are produced on the device
        and read back once."""
        eng = pl_module.engine
        eng.encoder_forward(pc, key=False, mode=L.MODE_INFER, patchesT=pc.patchesT, cls_tail=True)
        eng.heads_forward(pc, key=False, want_q=False)
        bc = eng.bt_bufs(n_real, "txtatk_cand")
        zc = eng.bt_forward(bc, pc.cls[:n_real].contiguous(), training=bool(pl_module.training), track=False)
        Z = ctx.clone()
        vals = torch.empty(n_real + 1, 2, device=Z.device)
        lam = pl_module.adv_lr
        eng.bt_loss_of(Z, self._zk, Bn, float(Bn), lam, 1.0, vals[0])
        for r in range(n_real):
            Z[owner[r]].copy_(zc[r])
            eng.bt_loss_of(Z, self._zk, Bn, float(Bn), lam, 1.0, vals[1 + r])
        v = vals.cpu().double()
        loss = (v[:, 0] + lam * v[:, 1]).tolist()
        out, r = [], 0
        for b in range(Bn):
            best, best_j, losses = loss[0], -1, []
            j = 0
            while r < n_real and owner[r] == b:
                losses.append(loss[1 + r])
                if loss[1 + r] > best:
                    best, best_j = loss[1 + r], j
                r += 1
                j += 1
            out.append((losses, best_j))
        return out
