"""History buffer of generated images for the temporal-coherence discriminator.
Same behaviour (incl. the use of Python's ``random``) as Module2/util/image_pool.py:5-54."""
import random

import torch


class ImagePool:
    def __init__(self, pool_size):
        self.pool_size = pool_size
        if self.pool_size > 0:
            self.num_imgs = 0
            self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for image in images:
            image = torch.unsqueeze(image.detach(), 0)
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image)
                out.append(image)
            elif random.uniform(0, 1) > 0.5:
                idx = random.randint(0, self.pool_size - 1)
                tmp = self.images[idx].clone()
                self.images[idx] = image
                out.append(tmp)
            else:
                out.append(image)
        return torch.cat(out, 0)
