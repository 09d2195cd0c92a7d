for c in 3 7; do echo "== cfg $c"; DM4D_CONV_CFG=$c python tools/conv_shapes.py 2>&1 | grep -E "UNet|VAE|sum over" | cut -c1-100; done
