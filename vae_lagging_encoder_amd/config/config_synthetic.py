"""Synthetic LSTM-generated corpus (reference config/config_synthetic.py): validation reuses the test split."""
from ._contract import lstm_text

_stem = "datasets/synthetic_data/synthetic_"
params = lstm_text("synthetic", nz=2, ni=50, nh=50, batch_size=16, epochs=50, test_nepoch=1,
                   split_names={"train_data": _stem + "train.txt", "val_data": _stem + "test.txt",
                                "test_data": _stem + "test.txt"})
