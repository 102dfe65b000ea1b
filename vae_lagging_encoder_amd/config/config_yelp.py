"""Yelp reviews (reference config/config_yelp.py): same shape as yahoo plus sentiment labels."""
from ._contract import lstm_text

params = lstm_text("yelp", nz=32, ni=512, nh=1024, batch_size=32, epochs=100, test_nepoch=5, label=True)
