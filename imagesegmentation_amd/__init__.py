"""MI355X-native hot paths of Shadowen/ImageSegmentation.

Two paths, behind the reference's own Python surface:
  * the ConvLSTM vertex-prediction recurrence  (convLSTM.ConvLSTMCell, polyrnn.Model)
  * polygon rasterisation + the IoU reward     (util.create_shape_mask, polyrl.PolygonEnv)

All arithmetic runs in the HIP kernels of libisg.so (see include/isg.h); importing this package
does not load the library, the first call does, and it raises if the library is missing.
"""
__version__ = "0.1.0"
