"""The overlay against the REAL reference checkout (build container only: skipped where /root/reference is absent, e.g. on
the GPU box).  What ``tests/test_dropin_cpu.py`` shows for imports on a stand-in tree is shown here for USE: the
reference's own losses / metrics run on the mirror's ``isegm.utils.misc``, the statements of
``scripts/evaluate_vpumodel.py`` that touch ``isegm.inference.utils`` work, a checkpoint written by ``save_checkpoint``
comes back through ``utils.load_is_model``, a config that pickles the reference's ``CrossEntropyLoss`` (what released
``.pth`` files contain) unpickles with the mirror alone, and every public name of every reference file that a
mirror module hides resolves (mirror's own definition or the fall-through of ``pvpuformer_amd._overlay``).

Third-party packages absent from the image are stood in for by ``oracle/ref_import.install_stubs()`` (test infrastructure).
Each case runs in a fresh interpreter so that ``sys.modules`` starts clean."""
import ast
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "isegm")), reason="reference checkout not present")

PRELUDE = textwrap.dedent(f"""
    import sys
    sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'oracle')!r})
    import ref_import
    ref_import.install_stubs()
    sys.path.remove({REF!r})                      # the overlay is told where the reference is, it is not first on sys.path
    import pvpuformer_amd
    pkg = pvpuformer_amd.install(reference_root={REF!r})
    assert pkg.__vpu_overlay__ == {os.path.join(REF, 'isegm')!r}, pkg.__vpu_overlay__
    import numpy as np, torch
    """)


def _run(code, cwd, with_prelude=True):
    src = (PRELUDE if with_prelude else "") + textwrap.dedent(code)
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, cwd=str(cwd),
                       env=dict(os.environ, PYTHONPATH=""), timeout=900)
    assert r.returncode == 0 and "CASE-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-5000:]
    return r.stdout


def test_reference_losses_and_metrics_run_on_the_mirror_misc(tmp_path):
    _run("""
        from isegm.model.losses import NormalizedFocalLossSigmoid, SigmoidBinaryCrossEntropyLoss, DiceLoss
        from isegm.model.metrics import AdaptiveIoU
        import isegm.model.losses, isegm.utils.misc
        assert isegm.model.losses.__file__.startswith('/root/reference')            # the reference's own file ...
        assert isegm.model.losses.misc is isegm.utils.misc                          # ... on the mirror's helper module
        assert isegm.utils.misc.__name__ == 'pvpuformer_amd.isegm.utils.misc'
        g = torch.Generator().manual_seed(0)
        x = torch.randn(2, 1, 32, 32, generator=g); y = (torch.rand(2, 1, 32, 32, generator=g) > 0.5).float()
        nfl = NormalizedFocalLossSigmoid(alpha=0.5, gamma=2)(x, y)
        bce = SigmoidBinaryCrossEntropyLoss()(x, y)
        assert nfl.shape == (2,) and bce.shape == (2,) and torch.isfinite(nfl).all() and torch.isfinite(bce).all()
        # against the closed forms (losses.py:155-176: mean over all but the batch axis)
        want = torch.nn.functional.binary_cross_entropy_with_logits(x, y, reduction='none').mean(dim=(1, 2, 3))
        assert torch.allclose(bce, want, atol=1e-6), (bce, want)
        m = AdaptiveIoU()
        m.update(x, y)                                                               # metrics.py:90 get_dims_with_exclusion
        assert 0.0 <= m.get_epoch_value() <= 1.0
        print('CASE-OK')
        """, tmp_path)


def test_evaluate_script_statements_and_checkpoint_round_trip(tmp_path):
    _run("""
        from pathlib import Path
        # ---- the import block of scripts/evaluate_vpumodel.py:13-18
        from isegm.inference import utils
        from isegm.utils.exp import load_config_file
        from isegm.utils.vis import draw_probmap, draw_with_blend_and_clicks, draw_with_blend_and_prompts, draw_with_error
        from isegm.inference.predictors import get_predictor
        from isegm.inference.vpu_evaluation import evaluate_dataset
        from isegm.model.modeling.pos_embed import interpolate_pos_embed_inference
        import isegm.utils.exp
        assert isegm.utils.exp.__file__.startswith('/root/reference')
        # ---- :114 get_dataset (isegm.data is absent from the reference snapshot: unknown -> None, known -> needs that package)
        assert utils.get_dataset('NoSuchSet', {}) is None
        try:
            utils.get_dataset('GrabCut', {'GRABCUT_PATH': '/nowhere'})
        except ModuleNotFoundError as e:
            assert 'isegm.data' in str(e)
        # ---- a model of the mirror, saved the way trainer.py:257-264 saves it, found and loaded the way :234 / :118 do
        from isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model
        from isegm.utils.misc import save_checkpoint
        bp = dict(img_size=(448, 448), patch_size=(16, 16), in_chans=3, embed_dim=128, depth=8, num_heads=4, mlp_ratio=4, qkv_bias=True)
        model = VitMultiGaussianVector_ed_Model(
            use_disks=True, norm_radius=5, with_prev_mask=True, with_aux_output=True, backbone_params=bp,
            neck_params=dict(in_dim=128, out_dims=[16, 32, 64, 128], img_size=(448, 448)),
            head_params=dict(in_channels=[16, 32, 64, 128], in_index=[0, 1, 2, 3], dropout_ratio=0.1, num_classes=1,
                             loss_decode=None, align_corners=False, upsample='x1', ed_loss=True, channels=32),
            random_split=False, residual=True)
        weights = Path('weights/toy_vpu/checkpoints'); save_checkpoint(model, weights, epoch=7, verbose=True)
        found = utils.find_checkpoint('weights', 'toy:007')
        assert found.endswith('weights/toy_vpu/checkpoints/007.pth'), found
        assert utils.find_checkpoint('weights', 'toy_vpu/checkpoints/007.pth') == 'weights/toy_vpu/checkpoints/007.pth'
        again = utils.load_is_model(Path(found), 'cpu', False)
        assert type(again) is type(model) and not again.training
        assert all(not p.requires_grad for p in again.parameters())
        sd0, sd1 = model.state_dict(), again.state_dict()
        assert list(sd0) == list(sd1) and all(torch.equal(sd0[k].cpu(), sd1[k].cpu()) for k in sd0)
        both, models = utils.load_is_model([torch.load(found, weights_only=False)] * 2, 'cpu', False)
        assert len(models) == 2 and type(both) is type(model)
        from isegm.utils.serialization import get_config_repr
        assert get_config_repr(model._config).startswith('Model: isegm.model.is_vpu_model.VitMultiGaussianVector_ed_Model')
        # ---- :253-264 the results table
        ious = [np.array([0.5, 0.82, 0.91, 0.96]), np.array([0.3, 0.4, 0.86, 0.86, 0.97])]
        spc, spi = utils.get_time_metrics(ious, 18.0)
        assert spc == 2.0 and spi == 9.0
        noc, std, over = utils.compute_noc_metric(ious, [0.8, 0.85, 0.9, 0.95], max_clicks=20)
        assert noc == [2.5, 3.0, 4.0, 4.5] and over == [0, 0, 0, 0]
        header, row = utils.get_results_table(noc, over, 'NoBRS', 'GrabCut', spc, 18.0, 20, model_name='toy')
        lines = header.split(chr(10))
        assert lines[0] == 'Eval results for model: toy' and len(lines[2]) == len(row) == len(lines[1]) == len(lines[3])
        assert row.split('|')[1:6] == ['    NoBRS    ', '  GrabCut  ', '  2.50   ', '  3.00   ', '  4.00   ']
        assert row.split('|')[-3:-1] == [' 2.000 ', ' 0:00:18 '], row
        assert '>=20@85%' in lines[2]
        _, short = utils.get_results_table(noc[:1], over[:1], 'NoBRS', 'DAVIS', spc, 18.0)
        assert short.count('?') == 6
        print('CASE-OK')
        """, tmp_path)


def test_released_checkpoint_config_unpickles_with_the_mirror_alone(tmp_path):
    # written by the REFERENCE's classes (its CrossEntropyLoss inside the model config, vpu_base448_cocolvis.py:39) ...
    _run(f"""
        import sys
        sys.path.insert(0, {os.path.join(ROOT, 'oracle')!r})
        import ref_import, torch
        ref_import.install_stubs()
        from isegm.model.modeling.transformer_helper.cross_entropy_loss import CrossEntropyLoss
        assert CrossEntropyLoss.__module__ == 'isegm.model.modeling.transformer_helper.cross_entropy_loss'
        import isegm
        assert isegm.__path__[0].startswith('/root/reference')
        loss = CrossEntropyLoss(use_sigmoid=False, loss_weight=1.0)
        head = dict(in_channels=[128, 256, 512, 1024], in_index=[0, 1, 2, 3], dropout_ratio=0.1, num_classes=1,
                    loss_decode=loss, align_corners=False, upsample='x1', ed_loss=True, channels=256)
        config = {{'class': 'isegm.model.is_vpu_model.VitMultiGaussianVector_ed_Model',
                  'params': {{'head_params': {{'type': 'builtin', 'value': head, 'specified': True}}}}}}
        torch.save({{'state_dict': {{}}, 'config': config}}, 'released.pth')
        print('CASE-OK')
        """, tmp_path, with_prelude=False)
    # ... and read back where neither the reference nor mmcv exists
    _run(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import pvpuformer_amd, torch
        pkg = pvpuformer_amd.install()
        assert pkg.__vpu_overlay__ is None and 'mmcv' not in sys.modules
        ck = torch.load('released.pth', map_location='cpu', weights_only=False)
        loss = ck['config']['params']['head_params']['value']['loss_decode']
        assert type(loss).__module__ == 'pvpuformer_amd.isegm.model.modeling.transformer_helper.cross_entropy_loss'
        assert loss.loss_weight == 1.0 and loss.reduction == 'mean' and loss.cls_criterion.__name__ == 'cross_entropy'
        x = torch.tensor([[2.0, 0.0], [0.0, 1.0]]); y = torch.tensor([0, 0])
        assert torch.allclose(loss(x, y), torch.nn.functional.cross_entropy(x, y))
        assert 'mmcv' not in sys.modules
        print('CASE-OK')
        """, tmp_path, with_prelude=False)


def _public_names(path):
    tree = ast.parse(open(path).read())
    names = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and not node.name.startswith("_"):
            names.append(node.name)
    return names


def _shadowed_pairs():
    """(dotted name below isegm, reference file) for every mirror module that hides a reference file."""
    mirror = os.path.join(ROOT, "pvpuformer_amd", "isegm")
    pairs = []
    for dirpath, _, files in os.walk(mirror):
        for f in files:
            if not f.endswith(".py"):
                continue
            rel = os.path.relpath(os.path.join(dirpath, f), mirror)
            ref_file = os.path.join(REF, "isegm", rel)
            if os.path.isfile(ref_file):
                dotted = rel[:-3].replace(os.sep, ".")
                dotted = dotted[:-len(".__init__")] if dotted.endswith(".__init__") else dotted
                pairs.append((dotted, ref_file))
    return sorted(pairs)


# modules whose hot-path surface the mirror must define ITSELF (no reliance on the fall-through): the three files round 2
# shadowed with a subset
OWN = ("utils.misc", "inference.utils", "utils.serialization")


def test_every_public_name_of_a_shadowed_reference_file_resolves(tmp_path):
    pairs = _shadowed_pairs()
    assert {"utils.misc", "inference.utils", "utils.serialization", "engine.trainer", "model.is_vpu_model"} <= {d for d, _ in pairs}
    table = {d: _public_names(p) for d, p in pairs if d != "__init__"}
    _run(f"""
        import importlib
        table = {table!r}
        own = {OWN!r}
        missing = []
        for dotted, names in table.items():
            mod = importlib.import_module('isegm.' + dotted)
            assert mod.__name__.startswith('pvpuformer_amd.'), mod
            for n in names:
                if dotted in own:
                    if n not in vars(mod):
                        missing.append((dotted, n, 'not defined by the mirror itself'))
                    continue
                try:
                    getattr(mod, n)
                except AttributeError as e:
                    missing.append((dotted, n, str(e)[:300]))
        assert not missing, missing
        # the fall-through serves a name only the reference file has, from the reference file
        import isegm.model.is_model
        f = isegm.model.is_model.split_points_by_order
        assert f.__code__.co_filename == '/root/reference/isegm/model/is_model.py', f.__code__.co_filename
        print('CASE-OK')
        """, tmp_path)
