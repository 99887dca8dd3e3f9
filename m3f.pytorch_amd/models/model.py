"""models.model -- the M3T valence-arousal task module, MI355X-native hot path.

Drop-in for the reference's models/model.py:27-493 (`AffWild2VA`): same constructor
(`hparams` namespace with the reference's flags, `add_model_specific_args`), same sub-module
attribute names (`visual`, `audio`, `proj_v`, `att_fuse`, `fusion`) and therefore the same
state_dict keys, same `forward(batch: dict)`, loss helpers and `training_step` return dict.

What runs where: every BiGRU / FC / fusion / loss op is a HIP kernel behind the C ABI
(include/m3t_hip.h).  Independent BiGRU stacks are advanced together (audio + the two visual
towers; the two fusion scorers) so that one launch per time step covers all of them.  The
3-D conv stems of the visual tower run on the library's tap-walk convolutions, BatchNorm and
pooling kernels too (models/backbone.py; rounds 4-6).  If pytorch_lightning is
installed the class derives from pl.LightningModule exactly as the reference does; without it
(this image) it is a plain nn.Module with the same hooks.

Validation/test window stitching is host glue in m3t/stitch.py.  Out of scope: dataloaders
(need the Aff-Wild2 dataset and cv2), the LR range finder, `--fusion_type att_dec`.
"""
import os
from argparse import ArgumentParser

import torch
import torch.nn as nn
from torch.nn import functional as F

from m3t import ops
from .backbone import VA_3DResNet, VA_3DVGGM, VA_3DVGGM_Split
from .rnn import GRU, run_grus, run_grus_cat
from .att_fusion import AttFusion
from .utils import concordance_cc2, mse  # noqa: F401  (re-exported like the reference)

try:  # pragma: no cover - pytorch_lightning is absent in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None
    _Base = nn.Module


# M3T_STEP_SYNC=1: training_step reads the loss statistics back after the loss (one host sync in the middle of the step, as until round 6)
# instead of deciding on the expression branch from a count that travels to the host under the forward pass (_count_valid_ahead).
_STEP_SYNC = os.environ.get("M3T_STEP_SYNC", "0") == "1"


class AffWild2VA(_Base):

    def __init__(self, hparams):
        super().__init__()
        try:
            self.hparams = hparams
        except AttributeError:      # newer Lightning: hparams is a read-only property
            self.save_hyperparameters(hparams)
        hp = hparams
        use_mtl = 'mtl' in hp.loss
        fc_outputs = 9 if use_mtl else 2          # 7 expression logits + valence + arousal (model.py:35)
        av = hp.modality == 'audiovisual'
        enc_classes = -1 if av else fc_outputs    # AV encoders emit raw 2H features (model.py:36-39)

        if 'visual' in hp.modality:
            common = dict(hiddenDim=hp.num_hidden, frameLen=hp.window, backend=hp.backend, nClasses=enc_classes,
                          nFCs=hp.num_fc_layers)
            if hp.backbone == 'resnet':
                self.visual = VA_3DResNet(resnet_ver='v1', **common)
            elif hp.backbone == 'v2p':
                self.visual = VA_3DVGGM(**common)
            elif hp.backbone == 'v2p_split':
                self.visual = VA_3DVGGM_Split(split_layer=hp.split_layer, use_mtl=use_mtl, **common)
            else:
                raise NotImplementedError("backbone '%s' is outside the MI355X hot path (SURVEY.md section 2.1)" % hp.backbone)
        if 'audio' in hp.modality:
            self.audio = GRU(200, 256, 2, enc_classes, hp.num_fc_layers)
        if av:
            self.proj_v = nn.Linear(hp.num_hidden * (2 if hp.split_layer == 5 else 4), 512)
            if hp.fusion_type == 'attention':
                self.att_fuse = AttFusion([512, 512], 128)
                self.fusion = GRU(512, hp.num_hidden, 2, fc_outputs, hp.num_fc_layers)
            elif hp.fusion_type == 'concat':
                self.fusion = GRU(512 * 2, hp.num_hidden, 2, fc_outputs, hp.num_fc_layers)
            else:
                raise NotImplementedError("fusion_type '%s' is outside the MI355X hot path" % hp.fusion_type)
        self.history = {'lr': [], 'loss': []}

    # ------------------------------------------------------------------ forward
    def _encode_av(self, batch, x):
        """audio encoder || visual towers: one grouped scan per layer when the visual back-end is the
        split GRU pair (the default), otherwise module by module."""
        se = batch['se_features']
        vis = self.visual
        if isinstance(vis, VA_3DVGGM_Split) and vis.split_layer != 5 and vis.backend == 'gru':
            x_v, x_a = vis.features(x, se, se)      # `se_features` passed twice, as model.py:111
            a, v12 = run_grus_cat([self.audio, vis.gru_v, vis.gru_a],
                                  [batch['audio'], ops.bct_to_btc(x_v), ops.bct_to_btc(x_a)], 1, 3)     # v12 = cat(v1, v2)
            return a, v12
        return self.audio(batch['audio']), vis(x, se, se)

    def forward(self, batch):
        hp = self.hparams
        if hp.modality == 'audio':
            return self.audio(batch['audio'])
        x = (batch['video'] - 127.5) / 127.5            # to [-1, 1] (model.py:106)
        if 'audio' in hp.modality:
            audio_feats, video_feats = self._encode_av(batch, x)
            video_feats = ops.linear(video_feats, self.proj_v.weight, self.proj_v.bias, 0)
            if hp.fusion_type == 'attention':
                return self.fusion(self.att_fuse(audio_feats, video_feats))
            return self.fusion(torch.cat((audio_feats, video_feats), dim=-1))
        return self.visual(x, batch['se_features'], batch['se_features'])

    # ------------------------------------------------------------------ losses (fused HIP kernel)
    def ccc_loss(self, y_hat, y):
        loss, _ = ops.va_loss(y_hat.reshape(-1, 1), y.reshape(-1), y.reshape(-1), iv=0, ia=0, w_v=1.0, w_a=0.0,
                              expr_w=0.0)
        return loss

    def mse_loss(self, y_hat, y):
        loss, _ = ops.va_loss(y_hat.reshape(-1, 1), y.reshape(-1), y.reshape(-1), iv=0, ia=0, w_v=1.0, w_a=0.0,
                              expr_w=0.0, use_mse=True)
        return loss

    def ce_loss(self, y_hat, y, mask):
        lg = y_hat.reshape(-1, y_hat.size(-1))
        zeros = lg.new_zeros(lg.size(0))
        loss, stats = ops.va_loss(lg, zeros, zeros, y.reshape(-1), mask.reshape(-1), iv=0, ia=0,
                                  n_expr=lg.size(-1), w_v=0.0, w_a=0.0, expr_w=1.0)
        return loss

    def bce_loss(self, y_hat, y, mask):     # AU branch: commented out upstream (model.py:184-199); host ops
        loss = F.binary_cross_entropy_with_logits(y_hat.view(-1), y.view(-1), reduction='none')
        return (loss * mask.view(-1).float()).mean()

    def va_objective(self, y_hat, batch):
        """Whole training objective in one kernel launch: returns (loss, stats) where stats =
        [loss, loss_v, loss_a, loss_expr, n_valid, n_correct, ccc_v, ccc_a] on the device."""
        hp = self.hparams
        mtl = 'mtl' in hp.loss
        if 'mse' not in hp.loss:
            assert 'ccc' in hp.loss, 'invalid loss specification'
        C_ = y_hat.size(-1)
        return ops.va_loss(y_hat, batch['label_valence'], batch['label_arousal'],
                           batch['class_expr'] if mtl else None, batch['expr_valid'] if mtl else None,
                           iv=7 if mtl else C_ - 2, ia=C_ - 1, n_expr=7 if mtl else 0,
                           w_v=hp.loss_lambda, w_a=1 - hp.loss_lambda, expr_w=0.8, use_mse='mse' in hp.loss)

    def _count_valid_ahead(self, mask):
        """valid_expr of the reference's training_step (models/model.py:173: `torch.sum(mask_expr_tile.long()).item()`) WITHOUT stalling the
        step: the count depends on the batch only, so it is queued before the forward pass and copied to pinned host memory while the host
        enqueues the forward pass; reading it afterwards waits for work that precedes this step, never for the forward pass -- the GPU keeps a
        full queue across the `if valid_expr > 0` decision (round 6: the read-back after the loss cost 1.3-2.3 ms of a 12.8-13.9 ms C5 step).
        Returns a callable giving the int."""
        if not mask.is_cuda:
            return lambda: int(mask.sum())
        buf = self.__dict__.get('_nvalid_host')
        if buf is None:
            buf = self.__dict__['_nvalid_host'] = torch.empty(1, dtype=torch.int64, pin_memory=True)
        buf.copy_(mask.sum(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()

        def get():
            ev.synchronize()
            return int(buf[0])
        return get

    def training_step(self, batch, batch_idx):
        mtl = 'mtl' in self.hparams.loss
        ahead = mtl and not _STEP_SYNC
        n_valid = self._count_valid_ahead(batch['expr_valid']) if ahead else None
        y_hat = self.forward(batch)
        loss, stats = self.va_objective(y_hat, batch)
        loss_v, loss_a = stats[1], stats[2]
        progress = {'loss_v': loss_v, 'loss_a': loss_a, 'loss': loss}
        log = {'loss_v': loss_v, 'loss_a': loss_a, 'loss': loss}
        if ahead:
            # acc_expr is a 0-dim device tensor like its neighbours in the dict (the reference: a Python float, two .item() syncs at
            # model.py:173,181); float(acc_expr) / Lightning's progress bar read it when they need it.  M3T_STEP_SYNC=1: the float.
            if n_valid() > 0:
                log['loss_expr'] = progress['loss_expr'] = stats[3]
                progress['acc_expr'] = stats[5] / stats[4]
        elif mtl:
            s = stats.tolist()      # ONE host sync after the loss (the reference has two .item() syncs, model.py:173,181)
            if s[4] > 0:
                log['loss_expr'] = progress['loss_expr'] = stats[3]
                progress['acc_expr'] = s[5] / s[4]
        if getattr(self.hparams, 'test_lr', False):
            raise NotImplementedError("LR range finder (models/lr_finder.py) is out of scope")
        return {'loss': loss, 'progress_bar': progress, 'log': log}

    # ------------------------------------------------------------------ evaluation glue (host ops)
    def _window_outputs(self, batch, with_gt):
        y_hat = self.forward(batch).detach().cpu()
        v_hat, a_hat = y_hat[..., -2], y_hat[..., -1]
        lens = batch['length']
        n = lens.size(0)
        out = {'v_pred': [v_hat[i][:lens[i]] for i in range(n)], 'a_pred': [a_hat[i][:lens[i]] for i in range(n)],
               'vid_names': batch['vid_name'], 'start_frames': batch['start'].cpu()}
        if with_gt:
            v, a = batch['label_valence'].cpu(), batch['label_arousal'].cpu()
            out['v_gt'] = [v[i][:lens[i]] for i in range(n)]
            out['a_gt'] = [a[i][:lens[i]] for i in range(n)]
        return out

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            return self._window_outputs(batch, True)

    def validation_end(self, outputs):
        from m3t import stitch
        m = stitch.val_metrics(outputs)
        gt_v, gt_a, pred_v, pred_a = stitch.stitch_val(outputs, self.hparams.window, self.hparams.test_on_val)
        torch.save({'valence_gt': gt_v, 'arousal_gt': gt_a, 'valence_pred': pred_v, 'arousal_pred': pred_a},
                   'predictions_val.pt')
        return {'val_loss': m['val_loss'],
                'progress_bar': {'val_ccc_v': m['val_ccc_v'], 'val_ccc_a': m['val_ccc_a']},
                'log': dict(m)}

    def test_step(self, batch, batch_idx):
        if self.hparams.test_on_val:
            return self.validation_step(batch, batch_idx)
        with torch.no_grad():
            return self._window_outputs(batch, False)

    def test_end(self, outputs):
        if self.hparams.test_on_val:
            return self.validation_end(outputs)
        from m3t import stitch
        pred_v, pred_a = stitch.stitch_test(outputs, self.hparams.window)
        torch.save({'valence_pred': pred_v, 'arousal_pred': pred_a}, 'predictions_test.pt')
        return {}

    def on_batch_end(self):
        if getattr(self.hparams, 'scheduler', None) == 'cyclic' and hasattr(self, 'cyclic_scheduler'):
            self.cyclic_scheduler.step()

    # ------------------------------------------------------------------ optimisation (host glue)
    def configure_optimizers(self):
        hp = self.hparams
        if hp.freeze_enc:
            for p in self.parameters():
                p.requires_grad = False
            live = [self.fusion, self.proj_v] + ([self.att_fuse] if hp.fusion_type == 'attention' else [])
            for m in live:
                for p in m.parameters():
                    p.requires_grad = True
        params = [p for p in self.parameters() if p.requires_grad]
        if hp.optimizer == 'adam':
            opt = torch.optim.Adam(params, lr=hp.learning_rate, weight_decay=1e-4)
        elif hp.optimizer == 'sgd':
            opt = torch.optim.SGD(params, lr=hp.learning_rate, momentum=0.9, weight_decay=5e-4)
        else:
            raise ValueError(hp.optimizer)
        if hp.scheduler == 'cyclic':
            self.cyclic_scheduler = torch.optim.lr_scheduler.CyclicLR(
                opt, hp.min_lr, hp.learning_rate, step_size_up=5000, cycle_momentum=hp.optimizer == 'sgd')
            return opt
        if hp.scheduler == 'exp':
            return [opt], [torch.optim.lr_scheduler.ExponentialLR(opt, hp.decay_factor)]
        if hp.scheduler == 'plateau':
            return [opt], [torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=hp.decay_factor, patience=3,
                                                                      min_lr=1e-6)]
        return opt

    @staticmethod
    def add_model_specific_args(parent_parser):
        """The reference's flags with the reference's defaults (model.py:448-493)."""
        parser = ArgumentParser(parents=[parent_parser])
        flags = [
            ('--backbone', 'v2p_split', str), ('--backend', 'gru', str), ('--modality', 'visual', str),
            ('--fusion_type', 'concat', str), ('--mode', 'video', str), ('--window', 32, int),
            ('--windows_per_epoch', 200, int), ('--learning_rate', 5e-5, float), ('--min_lr', 1e-8, float),
            ('--decay_factor', 0.5, float), ('--batch_size', 96, int), ('--optimizer', 'adam', str),
            ('--scheduler', 'plateau', str), ('--loss', 'ccc_mtl', str), ('--loss_lambda', 0.5, float),
            ('--num_hidden', 512, int), ('--split_layer', 3, int), ('--num_fc_layers', 2, int),
            ('--dataset_path', '/.data/zhangyuanhang/Aff-Wild2', str), ('--release', 'vipl', str),
            ('--input_size', 256, int), ('--checkpoint_path', '.', str), ('--workers', 8, int),
            ('--max_nb_epochs', 30, int),
        ]
        for name, default, typ in flags:
            parser.add_argument(name, default=default, type=typ)
        for name in ('--freeze_enc', '--resample', '--test_lr', '--test_on_val', '--cutout', '--distributed'):
            parser.add_argument(name, action='store_true', default=False)
        return parser
