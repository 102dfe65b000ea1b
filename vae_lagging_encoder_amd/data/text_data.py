"""Corpus -> vocabulary -> equal-length device batches, with the reference's observable behaviour (data/text_data.py).

What the hot path depends on:
  * ids: <pad> 0, <s> 1, </s> 2, <unk> 3, then words in first-seen order while the training corpus is read
    (text_data.py:75-110); a corpus read with an existing vocabulary maps unknown words to <unk>;
  * `create_data_batch` (text_data.py:219-255): sentences are ordered by length with `np.argsort` (numpy's default sort --
    the same call is used here, so sentences of equal length fall into the same batches in the same order), each run of
    equal length is cut into chunks of `batch_size`, every chunk becomes one int64 tensor [<s>, w_1 .. w_n, </s>] --
    all rows of a batch have the same length, so the LSTM kernels never see padding inside the aggressive loop;
  * `data_iter` (text_data.py:141-166): shuffled mini-batches of mixed length, longest first, padded with <pad>.
The batching itself is vectorised numpy (one array per length run) rather than per-token Python lists.
"""
import numpy as np
import torch

_SPECIALS = ("<pad>", "<s>", "</s>", "<unk>")


class VocabEntry(object):
    """word <-> id table (reference data/text_data.py:8-60)."""

    def __init__(self, word2id=None):
        if word2id:
            self.word2id = dict(word2id)
        else:
            self.word2id = {w: i for i, w in enumerate(_SPECIALS)}
        self.unk_id = self.word2id["<unk>"]
        self.id2word_ = {i: w for w, i in self.word2id.items()}

    def __getitem__(self, word):
        return self.word2id.get(word, self.unk_id)

    def __contains__(self, word):
        return word in self.word2id

    def __len__(self):
        return len(self.word2id)

    def add(self, word):
        wid = self.word2id.get(word)
        if wid is None:
            wid = self.word2id[word] = len(self.word2id)
            self.id2word_[wid] = word
        return wid

    def id2word(self, wid):
        return self.id2word_[wid]

    def decode_sentence(self, sentence):
        return [self.id2word_[int(w)] for w in sentence]

    @staticmethod
    def from_corpus(fname):
        vocab = VocabEntry()
        with open(fname) as fin:
            for line in fin:
                for word in line.split():
                    vocab.add(word)
        return vocab


class MonoTextData(object):
    """A tokenised corpus, one sentence per line, optionally `label<TAB>sentence` (reference text_data.py:63-110)."""

    def __init__(self, fname, label=False, max_length=None, vocab=None):
        grow = vocab is None
        table = VocabEntry() if grow else vocab
        self.data, self.labels, self.dropped = [], ([] if label else None), 0
        with open(fname) as fin:
            for line in fin:
                if label:
                    # the reference reads field 0 as the label and field 1 ONLY as the sentence (text_data.py:89-91): further
                    # tab-separated columns are ignored, and a labelled line without a tab is an error there (IndexError), not a
                    # dropped sentence
                    fields = line.split("\t")
                    lb, words = fields[0], fields[1].split()
                else:
                    words = line.split()
                if not words or (max_length and len(words) > max_length):
                    self.dropped += 1
                    continue
                if label:
                    self.labels.append(lb)
                self.data.append([table.add(w) for w in words] if grow else [table[w] for w in words])
        self.vocab = table

    def __len__(self):
        return len(self.data)

    # ---- batching ----------------------------------------------------------------------------------------------------------
    def _frame(self, sents, batch_first, device):
        """sents: list of id lists -> int64 tensor with <s> in front, </s> behind, <pad> beyond a sentence's end;
        lengths count <s> and </s> (reference _to_tensor, text_data.py:112-139)."""
        lens = np.fromiter((len(s) for s in sents), dtype=np.int64, count=len(sents))
        width = int(lens.max()) + 2
        arr = np.full((len(sents), width), self.vocab["<pad>"], dtype=np.int64)
        arr[:, 0] = self.vocab["<s>"]
        for r, s in enumerate(sents):
            arr[r, 1:1 + len(s)] = s
            arr[r, 1 + len(s)] = self.vocab["</s>"]
        t = torch.from_numpy(arr if batch_first else np.ascontiguousarray(arr.T))
        return (t.to(device) if device is not None else t), (lens + 2).tolist()

    def _length_runs(self, batch_size):
        """Index chunks of the reference's bucketing: argsort by length, cut every equal-length run into batch_size pieces."""
        lens = np.array([len(s) for s in self.data])
        order = np.argsort(lens)
        sorted_len = lens[order]
        starts = np.flatnonzero(np.r_[True, sorted_len[1:] != sorted_len[:-1]])
        ends = np.r_[starts[1:], len(order)]
        for a, b in zip(starts, ends):
            for c in range(a, b, batch_size):
                yield order[c:min(c + batch_size, b)]

    def create_data_batch(self, batch_size, device, batch_first=False):
        """-> list of equal-length batches (reference text_data.py:219-255)."""
        out, total = [], 0
        for idx in self._length_runs(batch_size):
            t, _ = self._frame([self.data[i] for i in idx], batch_first, device)
            out.append(t)
            total += len(idx)
        assert total == len(self.data)
        return out

    def create_data_batch_labels(self, batch_size, device, batch_first=False):
        """-> (batches, labels per batch) (reference text_data.py:168-217)."""
        batches, labels = [], []
        for idx in self._length_runs(batch_size):
            t, _ = self._frame([self.data[i] for i in idx], batch_first, device)
            batches.append(t)
            labels.append([self.labels[i] for i in idx])
        return batches, labels

    def data_iter(self, batch_size, device, batch_first=False, shuffle=True):
        """Mini-batches of mixed length, longest sentence first (reference text_data.py:141-166); yields (tensor, lengths)."""
        index_arr = np.arange(len(self.data))
        if shuffle:
            np.random.shuffle(index_arr)
        batch_num = int(np.ceil(len(index_arr)) / float(batch_size))       # the reference's expression: a trailing partial batch is dropped
        for i in range(batch_num):
            ids = index_arr[i * batch_size:(i + 1) * batch_size]
            sents = sorted((self.data[j] for j in ids), key=lambda e: -len(e))
            yield self._frame(sents, batch_first, device)

    def data_sample(self, nsample, device, batch_first=False, shuffle=True):
        """A random subset framed like one data_iter batch (reference text_data.py:257-285)."""
        index_arr = np.arange(len(self.data))
        if shuffle:
            np.random.shuffle(index_arr)
        ids = index_arr[:nsample]
        sents = sorted((self.data[j] for j in ids), key=lambda e: -len(e))
        return self._frame(sents, batch_first, device)
