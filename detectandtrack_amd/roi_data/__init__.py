"""Host-side training data: anchor labelling, RoI sampling, keypoint targets (reference lib/roi_data/*.py).  These run on
the host in the reference too (data-loader threads / Python ops); the arrays they emit are the label blobs the loss
kernels read."""
