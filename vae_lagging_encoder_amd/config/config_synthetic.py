"""Hyper-parameter contract of the reference for dataset 'synthetic' (reference config/config_synthetic.py) -- pure data,
kept key-for-key so the `params` dict is interchangeable with the reference module of the same name."""

params = {'enc_type': 'lstm',
 'dec_type': 'lstm',
 'nz': 2,
 'ni': 50,
 'enc_nh': 50,
 'dec_nh': 50,
 'dec_dropout_in': 0.5,
 'dec_dropout_out': 0.5,
 'epochs': 50,
 'batch_size': 16,
 'test_nepoch': 1,
 'train_data': 'datasets/synthetic_data/synthetic_train.txt',
 'val_data': 'datasets/synthetic_data/synthetic_test.txt',
 'test_data': 'datasets/synthetic_data/synthetic_test.txt'}
