"""Linguistic host side of the greedy text attack (attack/greedy_attack_vilt.py:21-45 word filter,
:76-111 synonym table, :199-246 synonyms / importance / re-tokenisation, :346-360 word <-> sub-word map).

Everything here is plain host code over strings and numpy; the tensor work (saliency gradients, candidate scoring) stays
in librmcl_hip.so behind GreedyAttack_moco.  The resources the reference loads by NAME are injected or read from LOCAL
files only (there is no network here): a WordPiece tokenizer object, a stop-word collection, the counter-fitted word
vectors.  Nothing is downloaded; a missing resource is an error at construction, not a silent fallback."""
from __future__ import annotations

import os
import string
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

# Function words the attack never replaces (greedy_attack_vilt.py:21-45: the TextFooler list the reference embeds); the
# reference additionally unions nltk's English stop words, which the caller supplies (`stopwords=`) when it has them.
FUNCTION_WORDS = frozenset("""
a about above across after afterwards again against ain all almost alone along already also although am among amongst an
and another any anyhow anyone anything anyway anywhere are aren aren't around as at back been before beforehand behind
being below beside besides between beyond both but by can cannot could couldn couldn't d didn didn't doesn doesn't don
don't down due during either else elsewhere empty enough even ever everyone everything everywhere except first for former
formerly from hadn hadn't hasn hasn't haven haven't he hence her here hereafter hereby herein hereupon hers herself him
himself his how however hundred i if in indeed into is isn isn't it it's its itself just latter latterly least ll may me
meanwhile mightn mightn't mine more moreover most mostly must mustn mustn't my myself namely needn needn't neither never
nevertheless next no nobody none noone nor not nothing now nowhere o of off on once one only onto or other others
otherwise our ours ourselves out over per please s same shan shan't she she's should've shouldn shouldn't somehow
something sometime somewhere such t than that that'll the their theirs them themselves then thence there thereafter
thereby therefore therein thereupon these they this those through throughout thru thus to too toward towards under unless
until up upon used ve was wasn wasn't we were weren weren't what whatever when whence whenever where whereafter whereas
whereby wherein whereupon wherever whether which while whither who whoever whole whom whose why with within without won
won't would wouldn wouldn't y yet you you'd you'll you're you've your yours yourself yourselves
""".split())

SPECIAL_TOKENS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")


def load_tokenizer(spec):
    """``config["tokenizer"]``: a tokenizer OBJECT is used as is; a local directory holding ``vocab.txt`` (or the file
    itself) becomes a lower-casing WordPiece tokenizer; a bare hub name ("bert-base-uncased", the reference's
    greedy_attack_vilt.py:53) cannot be fetched here and returns None - the caller then has no linguistic side."""
    if spec is None or not isinstance(spec, (str, os.PathLike)):
        return spec
    path = os.fspath(spec)
    if os.path.isdir(path):
        path = os.path.join(path, "vocab.txt")
    if not os.path.isfile(path):
        return None
    from transformers import BertTokenizer
    with open(path, encoding="utf-8") as f:
        vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
    return BertTokenizer(vocab=vocab, do_lower_case=True)


def load_stopwords(spec) -> frozenset:
    """Stop words beside FUNCTION_WORDS (the reference: nltk's English list, greedy_attack_vilt.py:51): an iterable, or
    a local text file with one word per line / whitespace separated; None = none."""
    if spec is None:
        return frozenset()
    if isinstance(spec, (str, os.PathLike)):
        with open(os.fspath(spec), encoding="utf-8") as f:
            return frozenset(f.read().split())
    return frozenset(spec)


class WordFilter:
    """check_word (greedy_attack_vilt.py:246-249): True = do not attack this word."""

    def __init__(self, stopwords: Iterable[str] = ()):
        self.stopwords = frozenset(stopwords)

    def __call__(self, word: str) -> bool:
        # `word in string.punctuation` / `word in '...'` are SUBSTRING tests in the reference: the empty string and
        # '..' are filtered too
        return (word in SPECIAL_TOKENS or word in self.stopwords or word in string.punctuation or word in FUNCTION_WORDS
                or word in "...")


class SynonymTable:
    """Nearest neighbours in the counter-fitted embedding space (init_matrix, greedy_attack_vilt.py:76-111).

    ``id2word`` numbers the lines of the vector file, ``word2id`` maps a word to its (last) line; ``neighbours[idx]`` lists the words whose cosine
    similarity with word idx is among its top ``n_candidates`` (the word itself occupies one of those slots and is
    dropped, :101-106) and at least ``sim_thred``, most similar first; a word without such neighbours maps to itself.

    The reference materialises the full V x V cosine matrix (17 GB for the 65 713 counter-fitted words) and keeps each
    word's candidates in a Python ``set`` (iteration order depends on the process's string-hash seed).  Here the matrix is
    only ever formed in row blocks, and candidates are ordered by similarity - the same SETS, in a reproducible order.
    ``sim_path``: an optional precomputed matrix in .npy form, memory-mapped, never unpickled."""

    def __init__(self, embedding_path: str, n_candidates: int, sim_thred: float, sim_path: Optional[str] = None, block: int = 2048):
        self.word2id: Dict[str, int] = {}
        self.id2word: Dict[int, str] = {}
        rows: List[List[float]] = []
        with open(embedding_path, encoding="utf-8") as f:
            for line in f:
                parts = line.strip().split()
                rows.append([float(x) for x in parts[1:]])
                # one id per LINE (the reference's `word not in sim_id2word` tests a word against integer keys and is always
                # true, :86-88): a repeated word keeps the id of its LAST line
                self.id2word[len(self.id2word)] = parts[0]
                self.word2id[parts[0]] = len(self.id2word) - 1
        vec = np.asarray(rows, dtype=np.float64)
        vec = np.asarray(vec / np.linalg.norm(vec, axis=1, keepdims=True), dtype=np.float32)
        sim_all = None
        if sim_path and os.path.exists(sim_path):
            sim_all = np.load(sim_path, mmap_mode="r", allow_pickle=False)
        self.neighbours: Dict[int, List[str]] = {}
        V = len(self.id2word)
        k = min(n_candidates, vec.shape[0])
        for s in range(0, V, block):
            e = min(V, s + block)
            sim = np.asarray(sim_all[s:e]) if sim_all is not None else vec[s:e] @ vec.T
            top = torch.topk(torch.from_numpy(np.ascontiguousarray(sim)), k=k, dim=1)
            for r in range(e - s):
                idx = s + r
                out: List[str] = []
                for v, i in zip(top.values[r].tolist(), top.indices[r].tolist()):
                    if v < sim_thred:
                        break
                    if i == idx:
                        continue
                    if self.id2word[i] not in out:
                        out.append(self.id2word[i])
                self.neighbours[idx] = out or [self.id2word[idx]]

    def __contains__(self, word: str) -> bool:
        return word in self.word2id

    def __call__(self, word: str) -> List[str]:
        """get_synonym_by_cos (:199-203): an unknown word is its own only candidate."""
        idx = self.word2id.get(word)
        return [word] if idx is None else self.neighbours[idx]


def words_to_sub_words(tokenizer, words: Sequence[str], max_length: int) -> Dict[int, np.ndarray]:
    """calc_words_to_sub_words (:346-360) for one sentence: word index -> positions of its WordPiece tokens, counted from
    the first token AFTER [CLS]; words that would reach max_length are cut off (and everything behind them)."""
    out: Dict[int, np.ndarray] = {}
    position = 0
    for idx, word in enumerate(words):
        n = len(tokenizer.tokenize(word))
        if position + n >= max_length:
            break
        out[idx] = np.arange(position, position + n)
        position += n
    return out


def importance_scores(grads: np.ndarray, mapping: Dict[int, np.ndarray]) -> List[float]:
    """get_important_scores (:221-228): L1 norm of the MEAN saliency gradient over a word's sub-word tokens.
    grads [L-1, D]: rows of one sample without its [CLS] row."""
    scores = [0.0] * len(mapping)
    for i in range(len(mapping)):
        scores[i] = np.linalg.norm(np.mean(grads[mapping[i]], axis=0), ord=1)
    return scores


def encode_sentences(tokenizer, sentences: List[str], max_length: int):
    """get_inputs (:230-244): truncate / pad to max_length; returns (ids, attention mask) as int64 CPU tensors."""
    enc = tokenizer(sentences, truncation=True, padding="max_length", max_length=max_length, return_special_tokens_mask=True)
    return torch.tensor(enc["input_ids"], dtype=torch.int64), torch.tensor(enc["attention_mask"], dtype=torch.int64)


def decode_words(tokenizer, ids_row) -> List[str]:
    """adv_attack_samples :507-509: the sentence as the tokenizer spells it, split at single blanks."""
    return tokenizer.decode(ids_row, skip_special_tokens=True, clean_up_tokenization_spaces=False).split(" ")
