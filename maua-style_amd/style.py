"""Command-line entry point and workflow drivers with the reference's argument surface (reference style.py).

    python style.py --content C.png --style S.png [flags of config.py]            (img_img)
    torchrun --nproc-per-node 8 style.py --transfer_type vid_img --content frames/ --style S.png ...

`img_img` is the reference's coarse-to-fine loop (style.py:22-73): resume by file existence, bilinear rescaling of
content / styles / previous result, optional histogram matching before and after every scale, optim.optimize per
scale, PNG per scale.  `vid_img` is the reference's per-frame loop (style.py:145-311).  Without optical flow
(BASELINE config 4) the frames are independent optimisations that share the network and the style targets, so they
are block-partitioned over the ranks of a torchrun job (dist.py) - one RCCL broadcast of weights and style targets,
no per-iteration collective, every rank writes its own frames.  With a precomputed flow cache (`.flo` fields and
reliability PNGs under <output_dir>/flow, as the reference's flow networks leave them) the temporally consistent
loop runs: warped previous result as initialisation and as pixel-level temporal target (sequential, one rank).
`img_vid` (README of the reference: "not sure if this is actually working"; its driver fails in its own loader) follows the
reference's intended flow with clips as directories of frames: optim.optimize's '_vid' branch - sliding windows of B
frames with the cross-frame dynamic Gram term - is the part with reference parity (tests/golden/imgvid_S64.npz).
"""
import concurrent.futures
import glob
import math
import os
import os.path
import random

import numpy as np
import torch as th
from PIL import Image
import torch.nn.functional as F

import config
import dist
import load
import models
import optim
import plan
from utils import draw_match_noise, limit_host_threads, match_histogram, name


def _scaled_styles(style_images_big, content_area, args):
    out = []
    for img in style_images_big:
        scale = math.sqrt(content_area / (img.size(3) * img.size(2))) * args.style_scale
        out.append(_resize(th.clone(img), scale_factor=scale))
    return out


def _resize(img, size=None, scale_factor=None):
    """F.interpolate(mode="bilinear", align_corners=False) (reference style.py:38-66); device tensors take the HIP kernel."""
    if img.is_cuda:
        import hip
        return hip.resize_bilinear(img.float().contiguous(), size=size, scale_factor=scale_factor)
    if size is not None:
        return F.interpolate(img, size, mode="bilinear", align_corners=False)
    return F.interpolate(img, scale_factor=scale_factor, mode="bilinear", align_corners=False)


def _single_rank_only(what):
    """img_img / img_vid optimise ONE image / clip (replicas only): under a multi-rank launch rank 0
    does the job and the others leave, instead of every rank redoing it on GPU 0 and racing on the output files."""
    rank, _, world = dist.env_rank()
    if world > 1 and rank != 0:
        print(f"{what}: one job, rank {rank} of {world} has nothing to do")
        return False
    return True


def img_img(args):
    """Coarse-to-fine single image (reference style.py:22-73).  The content image, the style images and the pastiche are
    uploaded once and stay in HBM: colour matching (utils.match_histogram), the bilinear rescaling between scales and the
    optimisation itself all run on the device; the only transfers are the jitter draws of the colour matching (host RNG,
    upload) and the 8-bit image of every finished scale (download for the PNG, written by a background thread)."""
    if not _single_rank_only("img_img"):
        return None
    limit_host_threads()
    on_gpu = th.cuda.is_available()
    up = (lambda t: t.cuda()) if on_gpu else (lambda t: t)
    style_images_big = [up(t) for t in load.process_style_images(args)]
    content_image_big = match_histogram(up(load.preprocess(args.content)), style_images_big, mode=args.match_histograms)
    content_size = np.array(content_image_big.size()[-2:])
    pastiche = up(load.preprocess(args.init)) if args.init not in ("content", "random") else None

    writer, pending = concurrent.futures.ThreadPoolExecutor(max_workers=1), []
    for current_size, num_iters in zip(args.image_sizes, args.num_iters):
        print("\nCurrent size {}px".format(current_size))
        done = f"{args.output}_{current_size}.png"
        if os.path.exists(done):  # resume: a finished scale is reloaded instead of recomputed
            pastiche = up(load.preprocess(done))
            continue

        content_scale = current_size / max(*content_size)
        content_image = _resize(content_image_big, scale_factor=content_scale)
        style_images = _scaled_styles(style_images_big, content_image.shape[2] * content_image.shape[3], args)

        hw = tuple(int(v) for v in content_image.shape[2:])
        if args.init == "random" and pastiche is None:
            pastiche = up(th.randn(1, 3, *hw).mul(0.001))
        elif args.init == "content" and pastiche is None:
            pastiche = _resize(content_image_big.clone(), size=hw)
        else:
            pastiche = _resize(pastiche.clone(), size=hw)
        pastiche = match_histogram(pastiche, style_images_big, mode=args.match_histograms)

        output_image = optim.optimize(content_image, style_images, pastiche, num_iters, args, keep_on_device=on_gpu)

        pastiche = match_histogram(output_image.detach(), style_images_big, mode=args.match_histograms)
        # deprocessing = one kernel + a 3-byte-per-pixel download, here; PNG encoding (host only) runs beside the next scale
        pending.append(writer.submit(load.save_image_to_file, load.deprocess(pastiche.detach()), args,
                                     f"{args.output}_{current_size}"))
    for fut in pending:
        fut.result()
    writer.shutdown()
    return pastiche.cpu() if pastiche is not None else None


def img_vid(args):
    """One content image animated by style clips (reference style.py:76-142): the pastiche is a clip of `num_frames`
    frames (or as long as the longest style clip), optimised coarse-to-fine by optim.optimize's sliding windows of
    `gram_frame_window` frames (one value per scale).  As shipped the reference's own driver stops in
    load.process_style_videos (`args.style.split` on a list), so this follows its intended flow; clips are directories
    of frame images on both sides (no codecs in this build): style clips are read with load.preprocess_video, each
    scale's result goes to <output>_<size>/frame_#####.png and the final clip to <output>/."""
    import scipy.ndimage as ndi
    if not _single_rank_only("img_vid"):
        return None
    limit_host_threads()
    style_videos_big = load.process_style_videos(args)
    content_image_big = match_histogram(load.preprocess(args.content), style_videos_big, mode=args.match_histograms)
    video_length = max(v.shape[0] for v in style_videos_big) if args.num_frames == -1 else args.num_frames
    delta_ts = str(args.gram_frame_window).split(",")
    content_size = np.array(content_image_big.size()[-2:])
    H, W = (int(v) for v in content_size)

    if args.init == "random":
        pastiche = th.randn((video_length, 3, H, W)) * 255
        pastiche = th.from_numpy(ndi.gaussian_filter(pastiche.numpy(), [video_length, 0, H / 32, W / 32], mode="wrap"))
    elif args.init == "content":
        pastiche = F.interpolate(content_image_big.clone(), (H, W), mode="bilinear", align_corners=False)
        pastiche = pastiche.repeat([video_length, 1, 1, 1])
        pastiche += th.randn((video_length, 3, H, W)) * 255
        pastiche = th.from_numpy(ndi.gaussian_filter(pastiche.numpy(), [video_length, 0, 4, 4], mode="wrap"))
    else:
        pastiche = load.preprocess_video(args.init, args.fps).repeat([video_length, 1, 1, 1])
    pastiche = match_histogram(pastiche, style_videos_big, mode=args.match_histograms)

    for i, (current_size, num_iters) in enumerate(zip(args.image_sizes, args.num_iters)):
        done = f"{args.output}_{current_size}"
        if os.path.isdir(done) and os.listdir(done):  # resume: a finished scale is reloaded
            pastiche = load.preprocess_video(done, args.fps)
            continue
        print("\nCurrent size {}px".format(current_size))
        args.gram_frame_window = int(delta_ts[min(i, len(delta_ts) - 1)])
        content_image = F.interpolate(content_image_big, scale_factor=current_size / max(*content_size), mode="bilinear",
                                      align_corners=False)
        style_videos = _scaled_styles(style_videos_big, content_image.shape[2] * content_image.shape[3], args)
        pastiche = F.interpolate(pastiche.clone(), tuple(int(v) for v in content_image.shape[2:]), mode="bilinear",
                                 align_corners=False)

        pastiche = optim.optimize(content_image, style_videos, pastiche, num_iters, args).detach().cpu()

        # the reference rolls the clip and the style clips by 7 frames between scales (style.py:134-135), so that the
        # window seams fall elsewhere at the next scale
        pastiche = th.cat((pastiche[7:], pastiche[:7]))
        style_videos_big = [th.cat((svb[7:], svb[:7])) for svb in style_videos_big]
        if args.temporal_blend > 0:
            pastiche = th.from_numpy(ndi.gaussian_filter(pastiche.numpy(), [args.temporal_blend, 0, 0, 0], mode="wrap"))
        pastiche = match_histogram(pastiche, style_videos_big, mode=args.match_histograms)
        load.save_tensor_to_file(pastiche, args, filename=done)

    pastiche = match_histogram(pastiche, style_videos_big, mode=args.match_histograms)
    load.save_tensor_to_file(pastiche, args)
    return pastiche


def _vid_img_flow(args, output_dir, frames, style_images_big, content_size):
    """The reference's temporally consistent loop (style.py:153-305) over a PRECOMPUTED flow cache
    (`<output_dir>/flow/{forward,backward}_<prev>_<this>.flo` + `.png` reliability masks, as its flow networks leave them;
    estimating flow is outside this build).  Frames depend on their predecessor (warped previous result = initialisation
    and pixel-level temporal target), so this path is sequential: one rank runs it."""
    passes = max(1, args.passes_per_scale)
    prev_size = None
    for size_n, (current_size, num_iters) in enumerate(zip(args.image_sizes, args.num_iters)):
        nxt = args.image_sizes[min(len(args.image_sizes) - 1, size_n + 1)]
        if len(glob.glob("%s/%s/*.png" % (output_dir, nxt))) == len(frames):
            print("Skipping size: %s, already done." % current_size)
            prev_size = current_size
            continue
        print("\nCurrent size {}px".format(current_size))
        os.makedirs(output_dir + "/" + str(current_size), exist_ok=True)
        content_scale = current_size / max(*content_size)
        style_images = _scaled_styles(style_images_big, content_scale ** 2 * content_size[0] * content_size[1], args)
        optim.set_model_args(args, current_size)
        net, losses = models.load_model(args)

        for pass_n in range(passes):
            pastiche = None
            if args.loop:
                start_idx = random.randrange(0, len(frames) - 1)
                frames = frames[start_idx:] + frames[:start_idx]  # rotate frames
            if len(glob.glob("%s/%s/%s_*.png" % (output_dir, current_size, pass_n + 2))) == len(frames):
                print(f"Skipping pass: {pass_n + 1}, already done.")
                frames = list(reversed(frames))
                continue
            direction = "forward" if pass_n % 2 == 0 else "backward"
            pairs = zip(frames + frames[: 11 if args.loop else 1], frames[1:] + frames[: 10 if args.loop else 1])
            for n, (prev_frame, this_frame) in enumerate(pairs):
                args.output = "%s/%s/%s_%s.png" % (output_dir, current_size, pass_n + 1, name(this_frame))
                if os.path.isfile(args.output) and not n >= len(frames):
                    print("Skipping pass: %s, frame: %s. File already exists." % (pass_n + 1, name(this_frame)))
                    continue
                print("Optimizing... size: %s, pass: %s, frame: %s" % (current_size, pass_n + 1, name(this_frame)))
                content_frames = [
                    match_histogram(F.interpolate(load.preprocess(f), scale_factor=content_scale, mode="bilinear",
                                                  align_corners=False), style_images_big[0], mode=args.match_histograms)
                    for f in (prev_frame, this_frame)]
                stem = f"{output_dir}/flow/{direction}_{name(prev_frame)}_{name(this_frame)}"
                if size_n == 0 and pass_n == 0:
                    if args.init == "random":
                        pastiche = th.randn(content_frames[1].size()).mul(0.001)
                    elif args.init == "prev_warp":
                        if pastiche is None:
                            pastiche = content_frames[0]
                        flow_map = load.flow_warp_map(stem + ".flo", pastiche.shape[2:])
                        pastiche = F.grid_sample(pastiche, flow_map, padding_mode="border", align_corners=False)
                    else:
                        pastiche = content_frames[1].clone()
                else:
                    wrapped = n > len(frames)  # second lap of a looping clip reads this pass's own files
                    if pass_n == 0:  # last pass of the previous size
                        src_dir, src_pass = (current_size, pass_n + 1) if wrapped else (prev_size, passes)
                    else:            # previous pass of this size
                        src_dir, src_pass = current_size, (pass_n + 1 if wrapped else pass_n)
                    hw = content_frames[0].size()[2:]
                    if pastiche is None:
                        pastiche = load.preprocess("%s/%s/%s_%s.png" % (output_dir, src_dir, src_pass, name(prev_frame)))
                        if pass_n == 0:
                            pastiche = F.interpolate(pastiche, size=hw, mode="bilinear", align_corners=False)
                    blend_image = load.preprocess("%s/%s/%s_%s.png" % (output_dir, src_dir, src_pass, name(this_frame)))
                    if pass_n == 0:
                        blend_image = F.interpolate(blend_image, size=hw, mode="bilinear", align_corners=False)
                    flow_map = load.flow_warp_map(stem + ".flo", pastiche.shape[2:])
                    warp_image = F.grid_sample(pastiche, flow_map, padding_mode="border", align_corners=False)
                    reliable = F.interpolate(load.reliable_flow_weighting(stem + ".png"), size=pastiche.size()[2:],
                                             mode="bilinear", align_corners=False)
                    optim.set_temporal_targets(net, warp_image, warp_weights=reliable, args=args)
                    pastiche = (1 - args.temporal_blend) * blend_image + args.temporal_blend * pastiche

                out = optim.optimize(content_frames[1], style_images, pastiche, num_iters // passes, args, net, losses)
                pastiche = match_histogram(out.detach().cpu(), style_images_big[0], mode=args.match_histograms)
                disp = load.deprocess(pastiche.clone())
                if args.original_colors == 1:
                    disp = load.original_colors(load.deprocess(content_frames[1].clone()), disp)
                disp.save(str(args.output))
            frames = list(reversed(frames))  # the next pass runs the clip the other way
        prev_size = current_size
        del net
        th.cuda.empty_cache()


def _finish_frame(img, content_img, path, original_colors):
    """Writing one finished frame (reference style.py:294-297): optional colour transfer, PNG.  Host only (the deprocessing
    kernel already ran): safe on the background writer thread while the next batch optimises."""
    if original_colors == 1:
        img = load.original_colors(content_img, img)
    img.save(path)


def planned_frames(size):
    """How many independent frames of side `size` vid_img plans to evaluate per launch: enough pixels to fill the chip
    (4 x 1024^2), at most 16 frames.  This number fixes the convolutions' split-K policy for the whole job."""
    return max(1, min(16, (4 << 20) // max(1, int(size) * int(size))))


def frames_per_batch(size, args=None):
    """How many frames are actually optimised together: planned_frames, unless MAUA_FRAME_BATCH overrides it (1 = the
    reference's frame-by-frame loop; same results bit for bit, the policy above does not change).  Two flags make calls
    depend on each other or on their own file names and therefore run frame by frame: `--normalize_weights` divides the
    strengths of the SHARED network once per optimize call and never resets them (optim.py:176-178: they compound per frame),
    and `--save_iter` writes `<frame output>_<size>[_<iter>].png` from inside each frame's own call (optim.py:230-236)."""
    if args is not None and (getattr(args, "normalize_weights", False) or getattr(args, "save_iter", 0) > 0):
        return 1
    forced = plan.get_int("frame_batch")
    return forced if forced > 0 else planned_frames(size)


class _ByteBudget(dict):
    """dict of device tensors that forgets its oldest entries once it holds more than `budget` bytes (a miss is always
    recoverable: the frame is decoded / read from its file again).  Keeps vid_img's device memory O(1) in the clip length."""

    def __init__(self, budget):
        super().__init__()
        self.budget, self.held = int(budget), 0

    def __setitem__(self, key, t):
        if key in self:
            self.held -= super().__getitem__(key).numel() * super().__getitem__(key).element_size()
        super().__setitem__(key, t)
        self.held += t.numel() * t.element_size()
        while self.held > self.budget and len(self) > 1:
            oldest = next(iter(self))
            if oldest == key:
                break
            self.pop(oldest)

    def pop(self, key, *default):
        if key in self:
            t = super().pop(key)
            self.held -= t.numel() * t.element_size()
            return t
        if default:
            return default[0]
        raise KeyError(key)

    def clear(self):
        super().clear()
        self.held = 0


def _optimize_group(contents, style_images, inits, num_iters, args, net, losses, planned):
    """B frames at once through optim.optimize_frames; networks outside the fused plan go frame by frame."""
    import engine
    try:
        return optim.optimize_frames(th.cat(contents), style_images, th.cat(inits), num_iters, args, net, losses,
                                     planned_frames=planned)
    except engine.UnsupportedNet:
        return th.cat([optim.optimize(c, style_images, p, num_iters, args, net, losses, keep_on_device=True)
                       for c, p in zip(contents, inits)])


def vid_img(args):
    """Per-frame stylisation.  With a flow cache under <output_dir>/flow the reference's temporally consistent loop
    runs (sequential, rank 0); without one the frames are independent problems (reference style.py:192-290 minus flow): they
    are sharded over the ranks of the job, and every rank optimises its frames in batches (`frames_per_batch`) - the
    convolutions of a batch run together, each frame keeps its own losses and optimiser state, results are bit-identical to
    the frame-by-frame loop.  Frames stay on the device from decoding to the 8-bit image; the global RNG is drawn from in the
    frame-by-frame order (colour-matching jitter, random initialisation)."""
    limit_host_threads()
    rank, _, world = dist.init()
    output_dir = args.output_dir + "/" + name(args.content) + "_" + "_".join([name(s) for s in args.style])
    frames = load.process_content_frames(args.content)
    if args.temporal_weight > 0 and glob.glob(output_dir + "/flow/*.flo"):
        # sequential job (every frame starts from its predecessor): rank 0 runs it, the other ranks of a torchrun launch
        # leave at once - no collective here, a barrier would time out (and hold the GPUs) while rank 0 works for hours
        if rank == 0:
            _vid_img_flow(args, output_dir, frames, load.process_style_images(args),
                          np.array(load.preprocess(frames[0]).size()[-2:]))
        return
    lo, hi = dist.shard_range(len(frames), rank, world)
    mine = frames[lo:hi]
    content_size = np.array(load.preprocess(frames[0]).size()[-2:])
    on_gpu = th.cuda.is_available()
    up = (lambda t: t.cuda()) if on_gpu else (lambda t: t)
    style_images_big = [up(t) for t in load.process_style_images(args)]
    passes = max(1, args.passes_per_scale)
    mode = args.match_histograms

    prev_size = None
    writer, pending = concurrent.futures.ThreadPoolExecutor(max_workers=4), {}
    # What the next pass reads back is what this pass wrote: the 8-bit image of every finished frame stays in memory (768 KB at
    # 512 x 512), so the PNG only has to be decoded when it comes from an earlier run (resume) or was colour-transferred on
    # the way out.  Decoded, rescaled content frames are kept per scale as well (every pass starts from them again).
    # Both are bounded (MAUA_FRAME_CACHE_MB, default 4096 MB each): a clip of thousands of frames must not grow the device
    # footprint with its length (the reference holds one frame at a time); a miss decodes the PNG again.
    budget = int(plan.get_float("frame_cache_mb") * (1 << 20))
    written, content_cache = _ByteBudget(budget), _ByteBudget(budget)
    replica = dist.ReplicaWeights()
    for size_n, (current_size, num_iters) in enumerate(zip(args.image_sizes, args.num_iters)):
        print("\nCurrent size {}px".format(current_size))
        os.makedirs(output_dir + "/" + str(current_size), exist_ok=True)
        content_cache.clear()
        content_scale = current_size / max(*content_size)
        content_area = content_scale ** 2 * content_size[0] * content_size[1]
        style_images = _scaled_styles(style_images_big, content_area, args)

        optim.set_model_args(args, current_size)
        net, losses = models.load_model(args)
        # (one broadcast per model and job: the first scale's, in the start-up phase; later scales copy the kept replica weights locally,
        #  so no rank waits in a collective for a rank that is hours behind; a --scaling_args table that changes the model between sizes
        #  gets one entry per model; nothing happens outside a process group)
        replica.sync(net, src=0, key=(str(getattr(args, "model_file", None)), str(getattr(args, "pooling", None))))
        batch = frames_per_batch(current_size, args)

        for pass_n in range(passes):
            out_path = lambda frame: "%s/%s/%s_%s.png" % (output_dir, current_size, pass_n + 1, name(frame))
            todo = []
            for frame in mine:
                if os.path.isfile(out_path(frame)):
                    print("Skipping pass: %s, frame: %s. File already exists." % (pass_n + 1, name(frame)))
                else:
                    todo.append(frame)
            for g0 in range(0, len(todo), batch):
                group = todo[g0:g0 + batch]
                contents, inits, post_noise = [], [], []
                for frame in group:  # host phase, frame by frame: every global-RNG draw in the reference's order
                    print("Optimizing... size: %s, pass: %s, frame: %s" % (current_size, pass_n + 1, name(frame)))
                    resized = content_cache.get(frame)
                    if resized is None:
                        resized = content_cache[frame] = _resize(up(load.preprocess(frame)), scale_factor=content_scale)
                    content = match_histogram(resized, style_images_big[0], mode=mode)
                    if size_n == 0 and pass_n == 0:
                        pastiche = up(th.randn(content.size()).mul(0.001)) if args.init == "random" else content.clone()
                    else:  # previous result of this frame: last pass of the previous size, or previous pass of this size
                        src = ("%s/%s/%s_%s.png" % (output_dir, prev_size, passes, name(frame)) if pass_n == 0 else
                               "%s/%s/%s_%s.png" % (output_dir, current_size, pass_n, name(frame)))
                        if src in written:  # this run wrote it: same bytes as the file holds
                            previous = load.preprocess_u8(written.pop(src))
                        else:
                            if src in pending:  # still being written by the background writer
                                pending.pop(src).result()
                            previous = up(load.preprocess(src))
                        pastiche = _resize(previous, size=tuple(int(v) for v in content.size()[2:]))
                    post_noise.append(draw_match_noise(content.shape, style_images_big[0], mode=mode))
                    contents.append(content)
                    inits.append(pastiche)
                args.output = out_path(group[0])
                outs = _optimize_group(contents, style_images, inits, num_iters // passes, args, net, losses,
                                       planned_frames(current_size))
                for k, frame in enumerate(group):
                    out = match_histogram(outs[k:k + 1], style_images_big[0], mode=mode, _noise=post_noise[k])
                    # deprocessing = one kernel + a 3-byte-per-pixel download, here; colour transfer and PNG encoding of this
                    # batch run on the writer threads beside the next batch's optimisation
                    if on_gpu and args.original_colors != 1:
                        import hip
                        u8 = hip.deprocess_u8(out.float().contiguous(), load._MEAN_BGR)
                        if not (size_n == len(args.image_sizes) - 1 and pass_n == passes - 1):  # nothing reads the last pass back
                            written[out_path(frame)] = u8
                        img = Image.fromarray(u8.cpu().numpy(), mode="RGB")
                    else:
                        img = load.deprocess(out)
                    cimg = load.deprocess(contents[k]) if args.original_colors == 1 else None
                    pending[out_path(frame)] = writer.submit(_finish_frame, img, cimg, out_path(frame), args.original_colors)
        prev_size = current_size
        del net
        if on_gpu:
            th.cuda.empty_cache()
    for fut in pending.values():  # surface any error of the background writer
        fut.result()
    writer.shutdown()
    dist.end_of_job_barrier()  # uneven shards of a long job: the one wait with the long timeout


if __name__ == "__main__":
    args = config.get_args()
    if args.seed >= 0:
        th.manual_seed(args.seed)
        if th.cuda.is_available():
            th.cuda.manual_seed_all(args.seed)
    {"img_img": img_img, "vid_img": vid_img, "img_vid": img_vid}[args.transfer_type](args)
