#include "scene.h"
namespace hrt_host {
void Scene::setup_backrooms_pool() { clear(); error = "backrooms_pool: not built yet"; }
}
