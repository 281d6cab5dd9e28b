"""Checkpoint wire format and model dispatch: utils/checkpoints.py:13-49 of the reference.

A checkpoint is the reference's dict {'model': state_dict, 'optim': torch.optim.Adam state_dict, 'config': dict}
(+ optional meta keys), so files written here load in the reference and vice versa.  The optimizer of this build is
the fused flat Adam inside trainer.TrainStep; `AdamStateAdapter` presents its flat moments as / fills them from the
per-parameter layout torch.optim.Adam uses (state[i] = {step, exp_avg, exp_avg_sq}, i = position in
model.parameters()).
"""
from pathlib import Path
from typing import Any, Dict, Optional, Union

import torch

from .trainer import TrainStep


class AdamStateAdapter:
    """state_dict()/load_state_dict() of a torch.optim.Adam-shaped object over a TrainStep."""

    def __init__(self, train_step: TrainStep):
        self.ts = train_step

    def _views(self):
        f = self.ts.flat
        pos = {id(p): i for i, p in enumerate(f.params)}
        for p in self.ts.model.parameters():             # torch.optim.Adam indexes in model.parameters() order
            i = pos[id(p)]
            o, n = f.offsets[i], p.numel()
            yield p, self.ts.exp_avg[o:o + n].view_as(p), self.ts.exp_avg_sq[o:o + n].view_as(p)

    def state_dict(self) -> Dict[str, Any]:
        ts = self.ts
        state = {}
        n = 0
        for i, (p, m, v) in enumerate(self._views()):
            if ts.opt_step > 0:
                state[i] = {'step': torch.tensor(float(ts.opt_step)), 'exp_avg': m.clone(), 'exp_avg_sq': v.clone()}
            n += 1
        group = {'lr': ts.lr, 'betas': tuple(ts.betas), 'eps': ts.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(n))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        ts = self.ts
        groups = sd['param_groups']
        if len(groups) != 1:
            raise ValueError('expected a single Adam parameter group')
        g = groups[0]
        views = list(self._views())
        if len(g['params']) != len(views):
            raise ValueError(f"optimizer state has {len(g['params'])} parameters, the model {len(views)}")
        ts.lr = float(g['lr'])
        ts.betas = tuple(g['betas'])
        ts.eps = float(g['eps'])
        steps = set()
        for i, (p, m, v) in enumerate(views):
            st = sd['state'].get(g['params'][i])
            if st is None:
                m.zero_()
                v.zero_()
                continue
            m.copy_(st['exp_avg'].to(m.device))
            v.copy_(st['exp_avg_sq'].to(v.device))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError(f'per-parameter Adam step counts differ ({sorted(steps)}); the fused step keeps one')
        ts.opt_step = steps.pop() if steps else 0


def save_checkpoint(model: torch.nn.Module, optim, config: Dict[str, Any], path: Union[Path, str],
                    meta: Optional[Dict[str, Any]] = None) -> None:
    """utils/checkpoints.py:13-23.  `optim`: a torch optimizer, an AdamStateAdapter or a TrainStep."""
    if isinstance(optim, TrainStep):
        optim = AdamStateAdapter(optim)
    checkpoint = {'model': {k: v.detach().cpu() for k, v in model.state_dict().items()},
                  'optim': optim.state_dict(), 'config': config}
    if meta is not None:
        checkpoint.update(meta)
    torch.save(checkpoint, str(path))


def restore_checkpoint(model: torch.nn.Module, optim, path: Union[Path, str], device: torch.device) -> None:
    """utils/checkpoints.py:26-34 (silently does nothing when the file does not exist, like the reference).
    Loads with weights_only=True: nothing in the file is executed."""
    path = Path(path)
    if path.is_file():
        checkpoint = torch.load(path, map_location=device, weights_only=True)
        model.load_state_dict(checkpoint['model'])
        if isinstance(optim, TrainStep):
            optim = AdamStateAdapter(optim)
        optim.load_state_dict(checkpoint['optim'])
        print(f'Restored model with step {model.get_step()}\n')


def init_tts_model(config: Dict[str, Any]):
    """utils/checkpoints.py:37-49"""
    model_type = config.get('tts_model', 'forward_tacotron')
    if model_type == 'forward_tacotron':
        from .model import ForwardTacotron
        return ForwardTacotron.from_config(config)
    if model_type == 'fast_pitch':
        from .fastpitch import FastPitch
        return FastPitch.from_config(config)
    if model_type == 'multi_forward_tacotron':
        from .multi_model import MultiForwardTacotron
        return MultiForwardTacotron.from_config(config)
    if model_type == 'multi_fast_pitch':
        from .multi_fastpitch import MultiFastPitch
        return MultiFastPitch.from_config(config)
    raise ValueError(f'Model type not supported: {model_type}')
