"""Yahoo Answers (reference config/config_yahoo.py): the BASELINE.json headline corpus."""
from ._contract import lstm_text

params = lstm_text("yahoo", nz=32, ni=512, nh=1024, batch_size=32, epochs=100, test_nepoch=5)
