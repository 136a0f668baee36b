// SceneLoader.h — scene.xml + OBJ ingest with the reference's entry point
// (R/Scene/SceneLoader.h:9-12: static void LoadSceneFromXML(const std::string&, Scene*)).
#pragma once
#include <string>

#include "Scene.h"

namespace MetalCppPathTracer {

class SceneLoader {
public:
    // Same contract as the reference: on an unreadable file or a missing <Scene> root the scene is left
    // as the reference leaves it (untouched / cleared, R/Scene/SceneLoader.cpp:77-88) and a line is printed.
    static void LoadSceneFromXML(const std::string& path, Scene* scene);

    // Extension: status instead of printf, and a search root for mesh files.  The reference's bundled
    // scene.xml names its mesh by an absolute macOS path (R/scene.xml:16); a `file=` that cannot be opened
    // is retried as <assetRoot>/<basename> and then <directory of the XML>/<basename>.
    enum Status { Ok = 0, XmlUnreadable = 1, NoSceneRoot = 2, XmlMalformed = 3, MeshUnreadable = 4 };
    static Status Load(const std::string& path, Scene* scene, const std::string& assetRoot = std::string(),
                       std::string* log = nullptr);
};

}  // namespace MetalCppPathTracer
