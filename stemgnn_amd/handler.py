"""Device-side mirror of the reference driver ``models/handler.py``: same function names,
arguments and printed lines -- ``save_model, load_model, inference, validate, train, test`` -- with the batch gather,
the training step (hipGraph), the rolling inference, the de-normalisation and the metrics all on the GPU.

Differences a caller can see (all deliberate):
  * ``inference`` returns device tensors ([count, horizon, node] float32) instead of numpy arrays;
  * the per-step ``float(loss)`` host sync (handler.py:166) is replaced by a device accumulator read once per epoch;
  * ``--device cpu`` raises ``StemGNNHipError`` (no CPU path);
  * ``args.early_stop`` with no ``early_stop_step`` attribute falls back to 10 instead of raising AttributeError
    (handler.py:189 reads a flag main.py never defines).
"""
import json
import os
import time
from datetime import datetime

import numpy as np
import torch

from . import ops
from .base_model import Model
from .engine import TrainStep
from .forecast_dataloader import ForecastDataset, WindowLoader, denorm_coefficients
from .math_utils import Scores
from .optim import FusedRMSprop


def save_model(model, model_dir, epoch=None):
    """handler.py:16-24 (whole-module pickle; epoch 0 lands on the best-model name, as in the reference)."""
    if model_dir is None:
        return
    if not os.path.exists(model_dir):
        os.makedirs(model_dir)
    epoch = str(epoch) if epoch else ''
    file_name = os.path.join(model_dir, epoch + '_stemgnn.pt')
    with open(file_name, 'wb') as f:
        torch.save(model, f)


def load_model(model_dir, epoch=None):
    """handler.py:27-38."""
    if not model_dir:
        return
    epoch = str(epoch) if epoch else ''
    file_name = os.path.join(model_dir, epoch + '_stemgnn.pt')
    if not os.path.exists(model_dir):
        os.makedirs(model_dir)
    if not os.path.exists(file_name):
        return
    with open(file_name, 'rb') as f:
        model = torch.load(f, weights_only=False)
    return model


def inference(model, dataloader, device, node_cnt, window_size, horizon):
    """handler.py:41-65, every tensor staying on the GPU.  Returns (forecast [count,horizon,N], target) float32."""
    forecast_set, target_set = [], []
    model.eval()
    with torch.no_grad():
        for inputs, target in dataloader:
            inputs = inputs.to(device)
            target = target.to(device)
            step = 0
            forecast_steps = torch.zeros(inputs.shape[0], horizon, node_cnt, device=inputs.device)
            while step < horizon:
                forecast_result, _ = model(inputs)
                len_model_output = forecast_result.size()[1]
                if len_model_output == 0:
                    raise Exception('Get blank inference result')
                inputs = ops.roll_window(inputs, forecast_result, forecast_steps, step, horizon)
                step += min(horizon - step, len_model_output)
            forecast_set.append(forecast_steps)
            target_set.append(target)
    return torch.cat(forecast_set, dim=0), torch.cat(target_set, dim=0)


def validate(model, dataloader, device, normalize_method, statistic, node_cnt, window_size, horizon, result_file=None):
    """handler.py:68-100."""
    forecast_norm, target_norm = inference(model, dataloader, device, node_cnt, window_size, horizon)
    mul = add = None
    if normalize_method and statistic:
        mul, add = denorm_coefficients(normalize_method, statistic, forecast_norm.device)
    raw = Scores(target_norm, forecast_norm, mul, add)
    score, score_by_node = raw.get(), raw.get(by_node=True)
    score_norm = Scores(target_norm, forecast_norm).get() if mul is not None else score
    print(f'NORM: MAPE {score_norm[0]:7.9%}; MAE {score_norm[1]:7.9f}; RMSE {score_norm[2]:7.9f}.')
    print(f'RAW : MAPE {score[0]:7.9%}; MAE {score[1]:7.9f}; RMSE {score[2]:7.9f}.')
    if result_file:
        if not os.path.exists(result_file):
            os.makedirs(result_file)
        step_to_print = 0
        f2d, t2d = forecast_norm[:, step_to_print, :].double(), target_norm[:, step_to_print, :].double()
        if mul is not None:
            f2d, t2d = f2d * mul + add, t2d * mul + add
        forcasting_2d, forcasting_2d_target = f2d.cpu().numpy(), t2d.cpu().numpy()
        np.savetxt(f'{result_file}/target.csv', forcasting_2d_target, delimiter=",")
        np.savetxt(f'{result_file}/predict.csv', forcasting_2d, delimiter=",")
        np.savetxt(f'{result_file}/predict_abs_error.csv', np.abs(forcasting_2d - forcasting_2d_target), delimiter=",")
        with np.errstate(divide='ignore', invalid='ignore'):
            np.savetxt(f'{result_file}/predict_ape.csv',
                       np.abs((forcasting_2d - forcasting_2d_target) / forcasting_2d_target), delimiter=",")
    return dict(mae=score[1], mae_node=score_by_node[1], mape=score[0], mape_node=score_by_node[0],
                rmse=score[2], rmse_node=score_by_node[2])


def train(train_data, valid_data, args, result_file, model_factory=None, step_hook=None):
    """handler.py:103-192.  `model_factory(node_cnt, 2, window, multi, horizon=...)` replaces the Model constructor
    (tests pin dropout with it); `step_hook(epoch, i, train_step)` is called after every optimizer step."""
    node_cnt = train_data.shape[1]
    model = (model_factory or Model)(node_cnt, 2, args.window_size, args.multi_layer, horizon=args.horizon)
    model.to(args.device)
    if len(train_data) == 0:
        raise Exception('Cannot organize enough training data')
    if len(valid_data) == 0:
        raise Exception('Cannot organize enough validation data')

    if args.norm_method == 'z_score':
        normalize_statistic = {"mean": np.mean(train_data, axis=0).tolist(), "std": np.std(train_data, axis=0).tolist()}
    elif args.norm_method == 'min_max':
        normalize_statistic = {"min": np.min(train_data, axis=0).tolist(), "max": np.max(train_data, axis=0).tolist()}
    else:
        normalize_statistic = None
    if normalize_statistic is not None:
        with open(os.path.join(result_file, 'norm_stat.json'), 'w') as f:
            json.dump(normalize_statistic, f)

    if args.optimizer == 'RMSProp':
        my_optim = FusedRMSprop(model.parameters(), lr=args.lr, eps=1e-08)
    else:
        my_optim = torch.optim.Adam(params=model.parameters(), lr=args.lr, betas=(0.9, 0.999))
    my_lr_scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer=my_optim, gamma=args.decay_rate)

    train_set = ForecastDataset(train_data, window_size=args.window_size, horizon=args.horizon,
                                normalize_method=args.norm_method, norm_statistic=normalize_statistic,
                                device=args.device)
    valid_set = ForecastDataset(valid_data, window_size=args.window_size, horizon=args.horizon,
                                normalize_method=args.norm_method, norm_statistic=normalize_statistic,
                                device=args.device)
    train_loader = WindowLoader(train_set, batch_size=args.batch_size, drop_last=False, shuffle=True)
    valid_loader = WindowLoader(valid_set, batch_size=args.batch_size, shuffle=False)

    total_params = 0
    for name, parameter in model.named_parameters():
        if not parameter.requires_grad:
            continue
        total_params += parameter.numel()
    print(f"Total Trainable Params: {total_params}")

    stepper = TrainStep(model, my_optim, args.batch_size, args.window_size, args.horizon, node_cnt,
                        series=train_set.data, graph=getattr(args, "hipgraph", True))
    best_validate_mae = np.inf
    validate_score_non_decrease_count = 0
    performance_metrics = {}
    for epoch in range(args.epoch):
        epoch_start_time = time.time()
        model.train()
        cnt = 0
        for i, idx in enumerate(train_loader.index_batches()):
            stepper.run_indices(train_set.hi_all.index_select(0, idx))
            cnt += 1
            if step_hook is not None:
                step_hook(epoch, i, stepper)
        loss_total = stepper.epoch_loss_sum()
        ops.check_gather_status(train_set.device)
        print('| end of epoch {:3d} | time: {:5.2f}s | train_total_loss {:5.4f}'.format(epoch, (
                time.time() - epoch_start_time), loss_total / cnt))
        save_model(model, result_file, epoch)
        if (epoch + 1) % args.exponential_decay_step == 0:
            my_lr_scheduler.step()
        if (epoch + 1) % args.validate_freq == 0:
            is_best_for_now = False
            print('------ validate on data: VALIDATE ------')
            performance_metrics = validate(model, valid_loader, args.device, args.norm_method, normalize_statistic,
                                           node_cnt, args.window_size, args.horizon, result_file=result_file)
            if best_validate_mae > performance_metrics['mae']:
                best_validate_mae = performance_metrics['mae']
                is_best_for_now = True
                validate_score_non_decrease_count = 0
            else:
                validate_score_non_decrease_count += 1
            if is_best_for_now:
                save_model(model, result_file)
        if args.early_stop and validate_score_non_decrease_count >= getattr(args, "early_stop_step", 10):
            break
    return performance_metrics, normalize_statistic


def test(test_data, args, result_train_file, result_test_file):
    """handler.py:194-207."""
    with open(os.path.join(result_train_file, 'norm_stat.json'), 'r') as f:
        normalize_statistic = json.load(f)
    model = load_model(result_train_file)
    node_cnt = test_data.shape[1]
    test_set = ForecastDataset(test_data, window_size=args.window_size, horizon=args.horizon,
                               normalize_method=args.norm_method, norm_statistic=normalize_statistic,
                               device=args.device)
    test_loader = WindowLoader(test_set, batch_size=args.batch_size, drop_last=False, shuffle=False)
    performance_metrics = validate(model, test_loader, args.device, args.norm_method, normalize_statistic,
                                   node_cnt, args.window_size, args.horizon, result_file=result_test_file)
    mae, mape, rmse = performance_metrics['mae'], performance_metrics['mape'], performance_metrics['rmse']
    print('Performance on test set: MAPE: {:5.2f} | MAE: {:5.2f} | RMSE: {:5.4f}'.format(mape, mae, rmse))
    return performance_metrics
