"""Extracts the Dense weights of the reference's shipped MNIST baseline into a plain .npz.

Run ONCE in the build container with the interpreter that has h5py:
    /opt/conda/bin/python3.9 tests/golden/extract_mnist_weights.py
Source (data, not code): /root/reference/MNIST/nested_quantization_layer/baseline_model.keras
  -- a Keras-3 zip whose model.weights.h5 holds layers/dense/vars/{0:(784,128),1:(128,)} and
  layers/dense_1/vars/{0:(128,10),1:(10,)} (float32).  h5py executes nothing from the file.
Output: tests/golden/mnist_baseline_weights.npz  (W1, b1, W2, b2; float32, TF (in,out) layout).
Used as a realistic-distribution fixture for bit-exact integer tests.
"""
import io
import os
import zipfile

import h5py
import numpy as np

SRC = "/root/reference/MNIST/nested_quantization_layer/baseline_model.keras"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mnist_baseline_weights.npz")

with zipfile.ZipFile(SRC) as z:
    f = h5py.File(io.BytesIO(z.read("model.weights.h5")), "r")
    W1 = np.asarray(f["layers/dense/vars/0"], dtype=np.float32)
    b1 = np.asarray(f["layers/dense/vars/1"], dtype=np.float32)
    W2 = np.asarray(f["layers/dense_1/vars/0"], dtype=np.float32)
    b2 = np.asarray(f["layers/dense_1/vars/1"], dtype=np.float32)
assert W1.shape == (784, 128) and b1.shape == (128,) and W2.shape == (128, 10) and b2.shape == (10,)
np.savez_compressed(DST, W1=W1, b1=b1, W2=W2, b2=b2)
print("wrote", DST, os.path.getsize(DST), "bytes; max|W1| =", float(np.abs(W1).max()))
