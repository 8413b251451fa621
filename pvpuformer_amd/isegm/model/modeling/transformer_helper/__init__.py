"""Mirror stub of the reference's isegm/model/modeling/transformer_helper package (mmseg-derived helpers).

The VPU hot path uses none of its arithmetic (the segmentation head is re-implemented in ``pvpuformer_amd/engine.py``), but
released checkpoints pickle an instance of ``transformer_helper.cross_entropy_loss.CrossEntropyLoss`` inside
``config['params']['head_params']['value']['loss_decode']`` (models/iSegNet/vpu_base448_cocolvis.py:2,39), so that dotted
path must resolve for ``torch.load(..., weights_only=False)`` -- without mmcv.  Under the overlay every other name of the
package (``resize``, ``BaseDecodeHead`` ... used by the reference's swin_transformer.py:23) falls through to the
reference's own ``__init__`` (see ``pvpuformer_amd._overlay``)."""
