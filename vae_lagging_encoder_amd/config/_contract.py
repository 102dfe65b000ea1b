"""Builders for the per-dataset hyper-parameter dicts.

The reference keeps one `config/config_<dataset>.py` per corpus, each a literal dict that text.py:95-100 /
image.py:79-84 splat into argparse.  The key set is the contract (the trainers read `args.<key>`); here the
dicts are assembled from two small builders so every dataset file states only what differs.
"""


def _splits(stem, train="train", val="valid", test="test"):
    root = "datasets/%s_data/%s" % (stem, stem)
    return {"train_data": "%s%s.txt" % (root, train), "val_data": "%s%s.txt" % (root, val),
            "test_data": "%s%s.txt" % (root, test)}


def lstm_text(stem, nz, ni, nh, batch_size, epochs, test_nepoch, dropout=0.5, split_names=None, **extra):
    """LSTM encoder + LSTM decoder corpus (yahoo / yelp / synthetic)."""
    p = dict(enc_type="lstm", dec_type="lstm", nz=nz, ni=ni, enc_nh=nh, dec_nh=nh,
             dec_dropout_in=dropout, dec_dropout_out=dropout,
             batch_size=batch_size, epochs=epochs, test_nepoch=test_nepoch)
    p.update(split_names if split_names is not None else _splits(stem, ".train", ".valid", ".test"))
    p.update(extra)
    return p


def resnet_pixelcnn_image(stem, img_size, nz, enc_width, enc_blocks, dec_kernels, dec_width, latent_feature_map,
                          batch_size, epochs, test_nepoch):
    """ResNet encoder + gated PixelCNN decoder image set (omniglot): `dec_kernels` lists (kernel, repeat)."""
    ks = [k for k, rep in dec_kernels for _ in range(rep)]
    return dict(img_size=list(img_size), nz=nz, enc_layers=[enc_width] * enc_blocks, dec_kernel_size=ks,
                dec_layers=[dec_width] * len(ks), latent_feature_map=latent_feature_map, batch_size=batch_size,
                epochs=epochs, test_nepoch=test_nepoch, data_file="datasets/%s_data/%s.pt" % (stem, stem))
