"""The shipped network configuration (reference configs/inference/vista.yaml:20-40) and config plumbing."""

VISTA_UNET_KWARGS = dict(
    adm_in_channels=768, num_classes="sequential", use_checkpoint=False, in_channels=8, out_channels=4,
    model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
    num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1, context_dim=1024,
    spatial_transformer_attn_type="softmax-xformers", extra_ff_mix_layer=True, use_spatial_context=True,
    merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1], add_lora=False, action_control=True)


def unet_kwargs(model_channels=320, **over):
    kw = dict(VISTA_UNET_KWARGS)
    kw["model_channels"] = model_channels
    kw.update(over)
    return kw
