"""Synthetic text encoder shared by make_golden.py and the tests.  TEST INFRASTRUCTURE ONLY."""


def bank_encoder(bank, index, dtype=None):
    """Synthetic text encoder: label -> its row of the text bank, unknown text -> hash row."""
    def enc(text):
        key = text.replace(" ", "_")
        row = bank[index[key]] if key in index else bank[hash_row(key, bank.shape[0])]
        return row[None, :] if dtype is None else row[None, :].to(dtype)
    return enc


def hash_row(text, n):
    h = 2166136261
    for ch in text.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h % n
